// glue_capi.cpp — C entry points around glue_driver.hpp for tests/test_gpu_glue.py: the DuckDB glue
// (duckdb_shim/fit_agg_hip.cpp, compiled against the stand-in of DuckDB's headers) on top of the REAL library on a GPU,
// driven as a parallel hash aggregate and as a window aggregate.  Test infrastructure; builds into
// anofox-statistics_amd/duckdb_shim/libanofox_glue_capi.so (duckdb_shim/Makefile).
#include "glue_driver.hpp"
#include "family_driver.hpp"

using namespace glue_driver;

extern "C" {
#define GLUE_API __attribute__((visibility("default")))

static int fail(char *msg, const std::exception &e) {
	if (msg) {
		strncpy(msg, e.what(), 511);
		msg[511] = 0;
	}
	return -1;
}

GLUE_API void *glue_open(const char *fn_name, const char *options_spec /* NULL = no options argument */, int as_map, char *msg) {
	try {
		return new Query(fn_name, options_spec, as_map != 0);
	} catch (const std::exception &e) {
		fail(msg, e);
		return nullptr;
	}
}
GLUE_API void glue_close(void *q) { delete static_cast<Query *>(q); }
GLUE_API int glue_result_fields(void *q) { return (int)static_cast<Query *>(q)->ReturnType().children().size(); }
// arena statistics: [0] rows accepted [1] unrefined groups [2] slot high-water mark [3] live slots [4] fit calls [5] slots fitted
GLUE_API void glue_stats(void *q, int64_t *out6) {
	const auto a = static_cast<Query *>(q)->Stats();
	out6[0] = (int64_t)a.rows_accepted;
	out6[1] = a.unrefined;
	out6[2] = (int64_t)a.slot_high_water;
	out6[3] = (int64_t)a.live_slots;
	out6[4] = (int64_t)a.fit_calls;
	out6[5] = (int64_t)a.slots_fitted;
}

static Inputs make_inputs(size_t n, size_t p, const double *y, const double *x, const double *w, const uint8_t *y_null, const uint8_t *x_null,
                          const uint8_t *xe_null, const uint8_t *w_null) {
	Inputs in;
	in.n = n; in.p = p; in.y = y; in.x = x; in.w = w;
	in.y_null = y_null; in.x_null = x_null; in.xe_null = xe_null; in.w_null = w_null;
	return in;
}
static void copy_out(const Records &r, size_t rows, size_t p, double *core, double *inf, uint8_t *is_null) {
	for (size_t k = 0; k < rows; ++k) {
		is_null[k] = r.is_null[k];
		if (r.is_null[k] || r.p != p) continue;
		memcpy(core + k * (p + 6), &r.core[k * (p + 6)], (p + 6) * sizeof(double));
		if (inf && r.inference) memcpy(inf + k * (5 * p + 2), &r.inf[k * (5 * p + 2)], (5 * p + 2) * sizeof(double));
	}
}

// GROUP BY key: out_core [n_keys x (p + 6)] (last column = n_features), out_inf [n_keys x (5 p + 2)] or NULL, is_null [n_keys]
GLUE_API int glue_group_by(void *q, size_t n, size_t p, const uint32_t *key, size_t n_keys, const double *y, const double *x, const double *w,
                           const uint8_t *y_null, const uint8_t *x_null, const uint8_t *xe_null, const uint8_t *w_null, int n_threads,
                           size_t vector_size, int dictionary, double *out_core, double *out_inf, uint8_t *is_null, char *msg) {
	try {
		Records r = static_cast<Query *>(q)->GroupBy(make_inputs(n, p, y, x, w, y_null, x_null, xe_null, w_null), key, n_keys, n_threads, vector_size,
		                                             dictionary != 0);
		copy_out(r, n_keys, p, out_core, out_inf, is_null);
		return 0;
	} catch (const std::exception &e) {
		return fail(msg, e);
	}
}
// the same with a LIST length per row (x_len[r] <= p): groups of different widths in one query.  Row k of out_core keeps the
// p-wide layout; a group of q < p features has q coefficients, then NaN, and n_features = q in the last column
GLUE_API int glue_group_by_ragged(void *q, size_t n, size_t p, const uint32_t *x_len, const uint32_t *key, size_t n_keys, const double *y,
                                  const double *x, const double *w, int n_threads, size_t vector_size, double *out_core, double *out_inf,
                                  uint8_t *is_null, char *msg) {
	try {
		Inputs in = make_inputs(n, p, y, x, w, nullptr, nullptr, nullptr, nullptr);
		in.x_len = x_len;
		Records r = static_cast<Query *>(q)->GroupBy(in, key, n_keys, n_threads, vector_size, false);
		if (r.p > p) throw std::runtime_error("a result wider than the inputs");
		for (size_t k = 0; k < n_keys; ++k) {
			is_null[k] = r.is_null[k];
			if (r.is_null[k]) continue;
			const double *c = &r.core[k * (r.p + 6)];
			for (size_t j = 0; j < p; ++j) out_core[k * (p + 6) + j] = j < r.p ? c[j] : NAN;
			for (size_t j = 0; j < 6; ++j) out_core[k * (p + 6) + p + j] = c[r.p + j];
			if (out_inf && r.inference) {
				const double *f = &r.inf[k * (5 * r.p + 2)];
				for (size_t l = 0; l < 5; ++l)
					for (size_t j = 0; j < p; ++j) out_inf[k * (5 * p + 2) + l * p + j] = j < r.p ? f[l * r.p + j] : NAN;
				out_inf[k * (5 * p + 2) + 5 * p] = f[5 * r.p];
				out_inf[k * (5 * p + 2) + 5 * p + 1] = f[5 * r.p + 1];
			}
		}
		return 0;
	} catch (const std::exception &e) {
		return fail(msg, e);
	}
}
// the aggregate OVER (ROWS BETWEEN preceding PRECEDING AND CURRENT ROW), one output row per input row
GLUE_API int glue_window(void *q, size_t n, size_t p, const double *y, const double *x, const double *w, size_t preceding, size_t vector_size,
                         double *out_core, double *out_inf, uint8_t *is_null, char *msg) {
	try {
		Records r = static_cast<Query *>(q)->Window(make_inputs(n, p, y, x, w, nullptr, nullptr, nullptr, nullptr), preceding, vector_size);
		copy_out(r, n, p, out_core, out_inf, is_null);
		return 0;
	} catch (const std::exception &e) {
		return fail(msg, e);
	}
}
// a segment tree over leaves of `leaf` rows, frames of the `back` + 1 latest leaves: one output row per leaf
GLUE_API int glue_tree_window(void *q, size_t n, size_t p, const double *y, const double *x, const double *w, size_t leaf, size_t back,
                              size_t vector_size, double *out_core, double *out_inf, uint8_t *is_null, char *msg) {
	try {
		Records r = static_cast<Query *>(q)->TreeWindow(make_inputs(n, p, y, x, w, nullptr, nullptr, nullptr, nullptr), leaf, back, vector_size);
		copy_out(r, (n + leaf - 1) / leaf, p, out_core, out_inf, is_null);
		return 0;
	} catch (const std::exception &e) {
		return fail(msg, e);
	}
}

// ---- the rest of the family (duckdb_shim/family_agg_hip.cpp) ----
GLUE_API void *family_open(const char *fn_name, const char *options_spec, int as_map, int with_split, char *msg) {
	try {
		return new FamilyQuery(fn_name, options_spec, as_map != 0, with_split != 0);
	} catch (const std::exception &e) {
		fail(msg, e);
		return nullptr;
	}
}
GLUE_API void family_close(void *q) { delete static_cast<FamilyQuery *>(q); }
// 0 = LIST(STRUCT) of the predict aggregates, 1 = STRUCT of the window aggregates, 2 = LIST(DOUBLE) of vif_agg; fields = the STRUCT's
GLUE_API int family_result_shape(void *q, int *fields) {
	auto *fq = static_cast<FamilyQuery *>(q);
	const LogicalType &t = fq->ReturnType();
	if (fq->kind() == FamilyQuery::Kind::PREDICT_AGG) *fields = (int)t.children()[0].second.children().size();
	else if (fq->kind() == FamilyQuery::Kind::WINDOW) *fields = (int)t.children().size();
	else *fields = 0;
	return (int)fq->kind();
}
GLUE_API int family_registered(void *q, const char *name) { return (int)static_cast<FamilyQuery *>(q)->Loader().registered.count(name); }
// *_fit_predict_agg GROUP BY key: out_offsets [n_keys + 1], out_vals [n x 4] = {y, yhat, yhat_lower, yhat_upper}, out_flags [n]
// (1 y NULL, 2 / 4 / 8 yhat / lower / upper NULL, 16 is_training), is_null [n_keys]; returns the number of output rows
GLUE_API int64_t family_predict_group_by(void *q, size_t n, size_t p, const uint32_t *key, size_t n_keys, const double *y, const double *x,
                                         const double *w, const uint8_t *y_null, const uint8_t *x_null, const uint8_t *xe_null, const uint8_t *w_null,
                                         const uint8_t *split, int n_threads, size_t vector_size, int dictionary, int64_t *out_offsets,
                                         double *out_vals, uint8_t *out_flags, uint8_t *is_null, char *msg) {
	try {
		FamilyOut r = static_cast<FamilyQuery *>(q)->GroupBy(make_inputs(n, p, y, x, w, y_null, x_null, xe_null, w_null), split, key, n_keys, n_threads,
		                                                     vector_size, dictionary != 0);
		if (r.flags.size() > n) throw std::runtime_error("more output rows than input rows");
		memcpy(out_offsets, r.offsets.data(), (n_keys + 1) * sizeof(int64_t));
		memcpy(is_null, r.is_null.data(), n_keys);
		if (!r.flags.empty()) {
			memcpy(out_vals, r.vals.data(), r.vals.size() * sizeof(double));
			memcpy(out_flags, r.flags.data(), r.flags.size());
		}
		return (int64_t)r.flags.size();
	} catch (const std::exception &e) {
		return fail(msg, e);
	}
}
// vif_agg GROUP BY key: out [n_keys x p], is_null [n_keys]
GLUE_API int family_vif_group_by(void *q, size_t n, size_t p, const uint32_t *key, size_t n_keys, const double *x, const uint8_t *x_null,
                                 const uint8_t *xe_null, int n_threads, size_t vector_size, int dictionary, double *out, uint8_t *is_null, char *msg) {
	try {
		FamilyOut r = static_cast<FamilyQuery *>(q)->GroupBy(make_inputs(n, p, nullptr, x, nullptr, nullptr, x_null, xe_null, nullptr), nullptr, key, n_keys,
		                                                     n_threads, vector_size, dictionary != 0);
		for (size_t k = 0; k < n_keys; ++k) {
			is_null[k] = r.is_null[k];
			if (r.is_null[k]) continue;
			if ((size_t)(r.offsets[k + 1] - r.offsets[k]) != p) throw std::runtime_error("a VIF list of the wrong length");
			memcpy(out + k * p, &r.vals[(size_t)r.offsets[k]], p * sizeof(double));
		}
		return 0;
	} catch (const std::exception &e) {
		return fail(msg, e);
	}
}
// *_fit_predict OVER (ROWS BETWEEN preceding PRECEDING AND CURRENT ROW) (leaf = 0), or over a segment tree of `leaf`-row leaves and
// frames of back + 1 leaves (one output row per leaf): out [rows x 3], is_null [rows]
GLUE_API int family_window(void *q, size_t n, size_t p, const double *y, const double *x, const double *w, const uint8_t *y_null, const uint8_t *x_null,
                           const uint8_t *xe_null, const uint8_t *w_null, size_t preceding, size_t leaf, size_t back, size_t vector_size, double *out,
                           uint8_t *is_null, char *msg) {
	try {
		auto *fq = static_cast<FamilyQuery *>(q);
		const Inputs in = make_inputs(n, p, y, x, w, y_null, x_null, xe_null, w_null);
		FamilyOut r = leaf ? fq->TreeWindow(in, leaf, back, vector_size) : fq->Window(in, preceding, vector_size);
		memcpy(out, r.vals.data(), r.vals.size() * sizeof(double));
		memcpy(is_null, r.is_null.data(), r.is_null.size());
		return 0;
	} catch (const std::exception &e) {
		return fail(msg, e);
	}
}

} // extern "C"
