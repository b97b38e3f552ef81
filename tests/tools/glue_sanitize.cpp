// glue_sanitize.cpp — duckdb_shim/fit_agg_hip.cpp compiled against the DuckDB stand-in (duckdb_stub/) and a MOCK of the
// C ABI (mock_abi.hpp), under -Wall -Wextra with ASan / UBSan: registration (names, aliases, overloads), bind (option
// parsing, result type, Copy / Equals), and the Update / Combine / Finalize / Destroy protocol as a parallel hash
// aggregate, as the naive window aggregator and as a segment tree drive it (glue_driver.hpp).  The mock's "fit" is
// order-sensitive, so a row that reaches the wrong slot, or the right slot in the wrong order, changes the result.
// Test infrastructure only (tests/test_sanitizers_cpu.py); nothing here is shipped.
#include "glue_driver.hpp"
#include "family_driver.hpp"

#include "mock_abi.hpp"

#define CHECK(c)                                                           \
	do {                                                                   \
		if (!(c)) {                                                        \
			fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
			exit(1);                                                       \
		}                                                                  \
	} while (0)

using namespace glue_driver;

static std::string error_of(const char *fn, const char *spec, bool as_map = false) {
	try {
		Query q(fn, spec, as_map);
	} catch (const std::exception &e) {
		return e.what();
	}
	return "";
}

static void registration_and_bind() {
	Query q("anofox_stats_ols_fit_agg", nullptr, false);
	auto &reg = q.Loader().registered;
	const char *names[3] = {"anofox_stats_ols_fit_agg", "anofox_stats_ridge_fit_agg", "anofox_stats_wls_fit_agg"};
	const char *aliases[3] = {"ols_fit_agg", "ridge_fit_agg", "wls_fit_agg"};
	CHECK(reg.size() == 6);
	for (int k = 0; k < 3; ++k) {
		CHECK(reg.count(names[k]) && reg.count(aliases[k]));
		auto &info = reg.at(names[k]);
		auto &al = reg.at(aliases[k]);
		CHECK(info.alias_of.empty() && al.alias_of == names[k]);
		CHECK(info.on_conflict == OnCreateConflict::ALTER_ON_CONFLICT && al.on_conflict == OnCreateConflict::ALTER_ON_CONFLICT);
		CHECK(info.functions.functions.size() == 2 && al.functions.functions.size() == 2 && info.descriptions.size() == 2);
		const size_t base = k == 2 ? 3 : 2; // wls: (y, x, weight)
		auto &f0 = info.functions.functions[0], &f1 = info.functions.functions[1];
		CHECK(f0.arguments.size() == base && f1.arguments.size() == base + 1);
		CHECK(f0.arguments[0] == LogicalType(LogicalType::DOUBLE) && f0.arguments[1] == LogicalType::LIST(LogicalType::DOUBLE));
		CHECK(f1.arguments.back() == LogicalType(LogicalType::ANY) && f0.return_type == LogicalType(LogicalType::ANY));
		if (k == 2) CHECK(f0.arguments[2] == LogicalType(LogicalType::DOUBLE));
		CHECK(f0.bind && f0.destructor && f0.combine && f0.update && f0.finalize && f0.initialize && !f0.simple_update);
		CHECK(f0.state_size(f0) == 3 * sizeof(int64_t)); // slot, feature count, device shard
	}
	// defaults (ols_aggregate.cpp:48-52): 7 fields, intercept, no inference, 0.95, solver svd, no HC
	CHECK(q.ReturnType().children().size() == 7 && q.ReturnType().children()[0].first == "coefficients" && q.ReturnType().children()[6].first == "n_features");
	{
		const auto o = q.Stats().options;
		CHECK(o.model == ANOFOX_HIP_MODEL_OLS && o.fit_intercept && !o.compute_inference && o.confidence_level == 0.95 && o.solver == ANOFOX_SOLVER_SVD &&
		      o.hc_type == ANOFOX_HC_NONE);
	}
	// the options literal: STRUCT, aliases, case-insensitive keys, integers as booleans, unknown keys ignored
	{
		Query r("ols_fit_agg", "Intercept=0;INFERENCE=true;confidence=0.9;solver=QR;hc_type=HC3;full_output=true", false);
		const auto o = r.Stats().options;
		CHECK(!o.fit_intercept && o.compute_inference && o.confidence_level == 0.9 && o.solver == ANOFOX_SOLVER_QR && o.hc_type == ANOFOX_HC_HC3);
		CHECK(r.ReturnType().children().size() == 14 && r.ReturnType().children()[13].first == "f_pvalue" && r.Inference());
	}
	{ // ridge: alpha wins over lambda, lambda alone counts, scaling; a MAP literal; HC is not a ridge option
		Query r1("anofox_stats_ridge_fit_agg", "lambda=3.0;alpha=0.25;lambda_scaling=glmnet;hc_type=hc1", false);
		CHECK(r1.Stats().options.model == ANOFOX_HIP_MODEL_RIDGE && r1.Stats().options.alpha == 0.25 &&
		      r1.Stats().options.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET && r1.Stats().options.hc_type == ANOFOX_HC_NONE);
		Query r2("ridge_fit_agg", "lambda=3.0;fit_intercept=0.0", true);
		CHECK(r2.Stats().options.alpha == 3.0 && !r2.Stats().options.fit_intercept);
		Query r3("ridge_fit_agg", nullptr, false);
		CHECK(r3.Stats().options.alpha == 1.0 && r3.Stats().options.lambda_scaling == ANOFOX_LAMBDA_SCALING_RAW); // ridge_aggregate.cpp:49-54
		Query r4("anofox_stats_wls_fit_agg", "alpha=5;compute_inference=1", false);
		CHECK(r4.Stats().options.model == ANOFOX_HIP_MODEL_WLS && r4.Stats().options.compute_inference && r4.Stats().options.alpha == 0.0);
	}
	// the reference's error texts (map_options_parser.cpp:21-45,222-266)
	CHECK(error_of("ols_fit_agg", "solver=lu") == "Invalid solver: 'lu'. Valid values are 'qr', 'svd', 'cholesky'");
	CHECK(error_of("ols_fit_agg", "hc_type=hc9") == "Invalid hc_type: 'hc9'. Valid values are 'none', 'hc0', 'hc1', 'hc2', 'hc3'");
	CHECK(error_of("ridge_fit_agg", "lambda_scaling=auto") == "Invalid lambda_scaling: 'auto'. Valid values are 'raw', 'glmnet'");
	CHECK(error_of("ols_fit_agg", "fit_intercept=yes") == "Cannot convert value of type VARCHAR to boolean");
	CHECK(error_of("ols_fit_agg", "fit_intercept=null;solver=svd").empty()); // NULL entries are skipped
	{ // an options argument that does not fold to a constant is not parsed (ols_aggregate.cpp:348): defaults
		Query r("ols_fit_agg", "compute_inference=true;solver=lu", false, /*foldable=*/false);
		CHECK(r.ReturnType().children().size() == 7);
	}
	// Copy() shares the query's arena (every thread must reach the same device state); Equals compares options + arena
	{
		auto c = q.BindData().Copy();
		CHECK(HipAggSharedStateOf(*c) == HipAggSharedStateOf(q.BindData()) && c->Equals(q.BindData()) && q.BindData().Equals(*c));
		Query other("anofox_stats_ols_fit_agg", nullptr, false);
		CHECK(!other.BindData().Equals(q.BindData())); // another aggregate of the query: its own state
	}
}

// rows of the test data and what the mock must return for a list of rows
struct Data {
	size_t n, p;
	std::vector<double> y, x, w;
	std::vector<uint8_t> y_null, x_null, xe_null, w_null;
	std::vector<uint32_t> key;
	Inputs in() const {
		Inputs i;
		i.n = n; i.p = p; i.y = y.data(); i.x = x.data(); i.w = w.data();
		i.y_null = y_null.data(); i.x_null = x_null.data(); i.xe_null = xe_null.data(); i.w_null = w_null.data();
		return i;
	}
};
static Data make_data(size_t n, size_t p, size_t n_keys, unsigned seed, bool nulls) {
	Data d;
	d.n = n; d.p = p;
	unsigned long long rng = seed;
	auto next = [&] { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
	d.y.resize(n); d.x.resize(n * p); d.w.resize(n); d.key.resize(n);
	d.y_null.assign(n, 0); d.x_null.assign(n, 0); d.w_null.assign(n, 0); d.xe_null.assign(n * p, 0);
	for (size_t i = 0; i < n; ++i) {
		d.key[i] = next() % n_keys;
		d.y[i] = (double)(next() % 1000) / 7.0;
		for (size_t j = 0; j < p; ++j) d.x[i * p + j] = (double)(next() % 100) + (double)j;
		d.w[i] = 0.5 + (double)(next() % 4);
		if (nulls) {
			d.y_null[i] = next() % 11 == 0;
			d.x_null[i] = next() % 13 == 0;
			d.w_null[i] = next() % 17 == 0;
		}
	}
	return d;
}

static void group_by(const char *fn, size_t p, int threads, size_t vsize, bool dictionary, int shards = 1) {
	const bool weighted = std::string(fn).find("wls") != std::string::npos;
	const size_t n = 5000, K = 37;
	Data d = make_data(n, p, K, 99 + (unsigned)p, true);
	Query q(fn, nullptr, false);
	Records rec = q.GroupBy(d.in(), d.key.data(), K, threads, vsize, dictionary);
	CHECK(rec.is_null.size() == K);
	// expected order of a key's rows: thread 0's vectors in order, then thread 1's, ... (Combine appends the sources)
	for (size_t k = 0; k < K; ++k) {
		std::vector<Row> rows;
		for (int t = 0; t < threads; ++t) {
			size_t v = 0;
			for (size_t r0 = 0; r0 < n; r0 += vsize, ++v) {
				if ((int)(v % (size_t)threads) != t) continue;
				for (size_t r = r0; r < std::min(n, r0 + vsize); ++r) {
					if (d.key[r] != k || d.y_null[r] || d.x_null[r] || (weighted && d.w_null[r])) continue;
					rows.push_back(Row{d.y[r], d.x[r * p], weighted ? d.w[r] : 1.0});
				}
			}
		}
		std::vector<double> want(p + 6);
		mock_fit(rows, p, want.data());
		if (want[p + 5] != 0.0) { CHECK(rec.is_null[k]); continue; }
		CHECK(!rec.is_null[k] && rec.p == p);
		const double *c = &rec.core[k * (p + 6)];
		for (size_t j = 0; j < p + 5; ++j) CHECK(c[j] == want[j]);
		CHECK(c[p + 5] == (double)p); // n_features
	}
	CHECK(q.Stats().live_slots == 0);                                     // every state was destroyed
	// ONE batched fit for the whole GROUP BY — per device state when ANOFOX_HIP_DEVICES shards the query's states
	CHECK(q.Stats().fit_calls >= 1 && q.Stats().fit_calls <= (uint64_t)shards && q.Stats().slots_fitted >= K - 1);
	if (shards == 1) CHECK(q.Stats().fit_calls == 1);
}

// A prepared statement executed twice: DuckDB re-uses the bind data (and with it the query's device state).  Once the first
// execution's states have all been destroyed, the second one starts from a clean state — slot numbers from 0 again, the same
// results — instead of resting on what the first left behind.
static void prepared_twice() {
	const size_t n = 3000, K = 23, p = 3;
	Data d = make_data(n, p, K, 7, true);
	Query q("ols_fit_agg", nullptr, false);
	const int resets_before = g_reset_calls.load();
	Records r1 = q.GroupBy(d.in(), d.key.data(), K, 3, 128, false);
	const auto slots1 = q.Stats().slot_high_water;
	Records r2 = q.GroupBy(d.in(), d.key.data(), K, 3, 128, false);
	CHECK(g_reset_calls.load() == resets_before + 1);          // exactly one reset: at the second execution's first Initialize
	CHECK(q.Stats().slot_high_water == slots1);                  // the slot numbering started over
	CHECK(r1.is_null == r2.is_null && r1.core == r2.core);      // and the answers are the first execution's
	CHECK(q.Stats().live_slots == 0);
}

static void window_replay() {
	// test/sql/comprehensive_tests.test:425-444: y = 2 i + 1, x = i for i = 1..20, ROWS BETWEEN 4 PRECEDING AND CURRENT ROW
	Data d = make_data(20, 1, 1, 1, false);
	for (size_t i = 0; i < 20; ++i) { d.x[i] = (double)(i + 1); d.y[i] = 2.0 * (double)(i + 1) + 1.0; }
	for (size_t vsize : {(size_t)2048, (size_t)4}) {
		Query q("anofox_stats_ols_fit_agg", nullptr, false);
		Records rec = q.Window(d.in(), 4, vsize);
		CHECK(rec.is_null.size() == 20);
		int n5 = 0;
		for (size_t r = 0; r < 20; ++r) {
			const size_t frame = std::min<size_t>(r + 1, 5);
			if (frame < 2) { CHECK(rec.is_null[r]); continue; } // one row -> NULL (ols_aggregate.cpp:263-267)
			CHECK(!rec.is_null[r] && rec.core[r * 7 + 5] == (double)frame);
			if (rec.core[r * 7 + 5] == 5.0) ++n5;
			double sy = 0;
			for (size_t k = r + 1 - frame; k <= r; ++k) sy += d.y[k];
			CHECK(rec.core[r * 7] == sy); // the mock's first "coefficient": the frame's own rows
		}
		CHECK(n5 == 16);
		CHECK(q.Stats().live_slots == 0 && q.Stats().slots_fitted == 20); // every frame fitted once, nothing re-fitted
		if (vsize == 4) CHECK(q.Stats().slot_high_water <= 8);                 // slots of destroyed states are handed out again
	}
	// a segment tree: leaves of 4 rows, frames of 3 leaves, every leaf the source of up to 3 targets in one Combine call
	{
		Data t = make_data(64, 2, 1, 5, false);
		Query q("ols_fit_agg", nullptr, false);
		Records rec = q.TreeWindow(t.in(), 4, 2, 2048);
		CHECK(rec.is_null.size() == 16);
		for (size_t o = 0; o < 16; ++o) {
			std::vector<Row> rows;
			for (size_t l = o >= 2 ? o - 2 : 0; l <= o; ++l)
				for (size_t r = 4 * l; r < 4 * l + 4; ++r) rows.push_back(Row{t.y[r], t.x[r * 2], 1.0});
			double want[8];
			mock_fit(rows, 2, want);
			CHECK(!rec.is_null[o]);
			for (int j = 0; j < 7; ++j) CHECK(rec.core[o * 8 + j] == want[j]);
		}
		CHECK(q.Stats().live_slots == 0);
	}
}

static void errors_and_flags() {
	{ // LIST lengths that differ (ols_aggregate.cpp:165-175)
		Data d = make_data(10, 3, 1, 3, false);
		std::vector<uint32_t> len(10, 3);
		len[7] = 2;
		Inputs in = d.in();
		in.x_len = len.data();
		Query q("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q.GroupBy(in, d.key.data(), 1, 1, 2048, false);
		} catch (const std::exception &e) {
			msg = e.what();
		}
		CHECK(msg == "Inconsistent feature count: expected 3, got 2");
	}
	{ // the feature count is per STATE (:164-175): groups of one query may differ in width, each result has its own LIST lengths
		Data d = make_data(30, 4, 3, 11, false);
		std::vector<uint32_t> len(30);
		for (size_t i = 0; i < 30; ++i) len[i] = d.key[i] == 0 ? 4 : (d.key[i] == 1 ? 2 : 0); // group 2: empty x lists -> NULL
		Inputs in = d.in();
		in.x_len = len.data();
		Query q("ols_fit_agg", "compute_inference=true", false);
		Records rec = q.GroupBy(in, d.key.data(), 3, 2, 7, true);
		CHECK(rec.p == 4 && !rec.is_null[0] && !rec.is_null[1] && rec.is_null[2]);
		CHECK(rec.core[0 * 10 + 4 + 5] == 4.0 && rec.core[1 * 10 + 4 + 5] == 2.0 && std::isnan(rec.core[1 * 10 + 2])); // n_features per row
		size_t n0 = 0, n1 = 0;
		for (size_t i = 0; i < 30; ++i) (d.key[i] == 0 ? n0 : n1) += d.key[i] < 2;
		CHECK(rec.core[0 * 10 + 4 + 4] == (double)n0 && rec.core[1 * 10 + 4 + 4] == (double)n1);
		const auto st = q.Stats();
		CHECK(st.widths == 2 && st.live_slots == 0 && st.rows_accepted == n0 + n1);
		// a later row of another width in the SAME group is the reference's error, with the state's own count
		len[0] = d.key[0] == 0 ? 2 : 4;
		Query q2("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q2.GroupBy(in, d.key.data(), 3, 1, 2048, false);
		} catch (const std::exception &e) {
			msg = e.what();
		}
		CHECK(msg.rfind("Inconsistent feature count: expected ", 0) == 0);
		CHECK(q2.Stats().live_slots == 0);
	}
	{ // Combine of states with different feature counts (:217-220): two threads see one group at different widths
		Data d = make_data(8, 3, 1, 12, false);
		std::vector<uint32_t> len = {3, 3, 3, 3, 2, 2, 2, 2};
		Inputs in = d.in();
		in.x_len = len.data();
		Query q("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q.GroupBy(in, d.key.data(), 1, 2, 4, false); // vectors of 4 rows, dealt to 2 threads
		} catch (const std::exception &e) {
			msg = e.what();
		}
		CHECK(msg == "Cannot combine states with different feature counts: 3 vs 2" ||
		      msg == "Cannot combine states with different feature counts: 2 vs 3");
	}
	{ // NULL list elements become NaN (the fit's row filter drops such rows): they reach the state as rows
		Data d = make_data(6, 3, 1, 4, false);
		d.xe_null[3 * 3 + 0] = 1;
		Query q("ols_fit_agg", nullptr, false);
		Records rec = q.GroupBy(d.in(), d.key.data(), 1, 1, 2048, true);
		CHECK(!rec.is_null[0] && rec.core[3 + 4] == 6.0 && std::isnan(rec.core[2])); // mock: core[2] = sum x0 w -> NaN
	}
	{ // groups the device state flags as unrefined: SQL NULL, counted, and an error on request
		g_mock_unrefined_rows = 3; // the mock flags every group of exactly 3 rows
		Data d = make_data(9, 1, 3, 8, false);
		for (size_t i = 0; i < 9; ++i) d.key[i] = i < 3 ? 0 : (i < 7 ? 1 : 2); // 3, 4 and 2 rows
		Query q("ols_fit_agg", nullptr, false);
		Records rec = q.GroupBy(d.in(), d.key.data(), 3, 1, 2048, false);
		CHECK(rec.is_null[0] && !rec.is_null[1] && !rec.is_null[2] && q.Stats().unrefined == 1);
		setenv("ANOFOX_HIP_UNREFINED", "error", 1);
		Query q2("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q2.GroupBy(d.in(), d.key.data(), 3, 1, 2048, false);
		} catch (const std::exception &e) {
			msg = e.what();
		}
		unsetenv("ANOFOX_HIP_UNREFINED");
		CHECK(msg.find("1 group(s)") != std::string::npos);
		g_mock_unrefined_rows = -1;
	}
}

// ---- the rest of the family (duckdb_shim/family_agg_hip.cpp) ----
static std::string family_error_of(const char *fn, const char *spec, bool split = false) {
	try {
		FamilyQuery q(fn, spec, false, split);
	} catch (const std::exception &e) {
		return e.what();
	}
	return "";
}

static void family_registration() {
	FamilyQuery q("anofox_stats_ols_fit_predict_agg", nullptr, false, false);
	auto &reg = q.Loader().registered;
	CHECK(reg.size() == 3 * 4 + 3 * 2 + 2);
	for (const char *m : {"ols", "ridge", "wls"}) {
		const std::string model = m, primary = "anofox_stats_" + model + "_fit_predict_agg";
		const size_t base = model == "wls" ? 3 : 2;
		for (const std::string &name : {primary, model + "_fit_predict_agg", model + "_predict_agg", "anofox_stats_" + model + "_predict_agg"}) {
			CHECK(reg.count(name));
			auto &info = reg.at(name);
			CHECK(info.on_conflict == OnCreateConflict::ALTER_ON_CONFLICT && info.alias_of == (name == primary ? "" : primary));
			CHECK(info.functions.functions.size() == 4);
			auto &fs = info.functions.functions;
			CHECK(fs[0].arguments.size() == base && fs[1].arguments.size() == base + 1 && fs[2].arguments.size() == base + 1 && fs[3].arguments.size() == base + 2);
			CHECK(fs[1].arguments.back() == LogicalType(LogicalType::ANY) && fs[2].arguments.back() == LogicalType(LogicalType::VARCHAR));
			CHECK(fs[3].arguments[base] == LogicalType(LogicalType::VARCHAR) && fs[3].arguments.back() == LogicalType(LogicalType::ANY));
			for (auto &f : fs) CHECK(f.bind && f.destructor && f.combine && f.update && f.finalize && f.initialize && f.state_size(f) == sizeof(void *));
		}
		CHECK(reg.at(primary).descriptions.size() == 4);
		const std::string win = "anofox_stats_" + model + "_fit_predict";
		CHECK(reg.count(win) && reg.count(model + "_fit_predict") && reg.at(model + "_fit_predict").alias_of == win);
		CHECK(reg.at(win).functions.functions.size() == 2 && reg.at(win).descriptions.size() == 2);
		CHECK(reg.at(win).functions.functions[0].return_type.id() == LogicalTypeId::STRUCT); // the window aggregates name their type at registration
	}
	CHECK(reg.count("anofox_stats_vif_agg") && reg.at("vif_agg").alias_of == "anofox_stats_vif_agg");
	// result types
	auto &row = q.ReturnType().children()[0].second;
	CHECK(q.ReturnType().id() == LogicalTypeId::LIST && row.id() == LogicalTypeId::STRUCT && row.children().size() == 5);
	CHECK(row.children()[0].first == "y" && row.children()[3].first == "yhat_upper" && row.children()[4].second == LogicalType(LogicalType::BOOLEAN));
	FamilyQuery w("ridge_fit_predict", "alpha=0.5;fit_intercept=false", false, false);
	CHECK(w.ReturnType().children().size() == 3 && w.ReturnType().children()[1].first == "yhat_lower");
	FamilyQuery v("vif_agg", nullptr, false, false);
	CHECK(v.ReturnType() == LogicalType::LIST(LogicalType::DOUBLE));
	// option errors surface from bind with the reference's texts
	CHECK(family_error_of("ols_fit_predict_agg", "null_policy=keep").find("Invalid null_policy: 'keep'. Valid values are 'drop', 'drop_y_zero_x'") != std::string::npos);
	CHECK(family_error_of("wls_fit_predict_agg", "null_policy=DROP_Y_ZERO_X", true).empty());
	CHECK(family_error_of("ols_fit_predict", "solver=lu").find("Invalid solver") != std::string::npos);
	CHECK(error_of("ols_fit_agg", "null_policy=keep").find("Invalid null_policy") != std::string::npos); // every aggregate validates the key
}

// what the mock makes of a group's rows in the order the driver combines them (thread by thread, input order within a thread)
static void family_group_by(const char *fn, size_t p, int threads, size_t vsize, bool dictionary, bool with_split, const char *spec) {
	const std::string name = fn;
	const bool weighted = name.find("wls") != std::string::npos, ols = name.find("ols") != std::string::npos;
	const bool drop_zero = spec && std::string(spec).find("drop_y_zero_x") != std::string::npos;
	const size_t n = 4000, K = 29;
	Data d = make_data(n, p, K, 7 + (unsigned)p, true);
	std::vector<uint8_t> split(n);
	for (size_t i = 0; i < n; ++i) {
		split[i] = (uint8_t)((i * 2654435761u >> 7) % kSplitStringCount);
		if (i % 9 == 0) d.xe_null[i * p + (i % p)] = 1;   // NULL list elements
		if (i % 14 == 0) d.x[i * p + ((i / 14) % p)] = 0.0; // exact zeros for drop_y_zero_x
		if (weighted && i % 19 == 0) d.w[i] = i % 38 == 0 ? 0.0 : -1.0;
	}
	FamilyQuery q(fn, spec, false, with_split);
	const int before = g_predict_calls.load();
	FamilyOut out = q.GroupBy(d.in(), with_split ? split.data() : nullptr, d.key.data(), K, threads, vsize, dictionary);
	CHECK(g_predict_calls.load() - before == 1); // ONE batched call for the whole vector of states
	// the expected rows per key
	for (size_t k = 0; k < K; ++k) {
		std::vector<size_t> rows;
		for (int t = 0; t < threads; ++t)
			for (size_t i = 0; i < n; ++i)
				if (d.key[i] == k && (int)((i / vsize) % (size_t)threads) == t && !d.x_null[i] && !(weighted && d.w_null[i])) rows.push_back(i);
		std::vector<uint8_t> train(rows.size());
		double S = 0.0;
		int64_t n_train = 0, counted = 0;
		for (size_t r = 0; r < rows.size(); ++r) {
			const size_t i = rows[r];
			bool tr = !d.y_null[i];
			if (with_split) {
				const char *sv = kSplitStrings[split[i]];
				std::string low = sv ? sv : "";
				for (auto &c : low) c = (char)tolower(c);
				tr = tr && (low == "train" || low == "training");
			}
			bool null_feature = false, zero = false;
			for (size_t j = 0; j < p; ++j) {
				null_feature = null_feature || d.xe_null[i * p + j];
				zero = zero || (!d.xe_null[i * p + j] && d.x[i * p + j] == 0.0);
			}
			if (weighted && !(d.w[i] > 0)) tr = false;
			if (null_feature && ols) tr = false;
			if (tr && drop_zero && zero) tr = false;
			train[r] = tr;
			counted += tr;
			// the library's row filter: a training row with a NaN feature does not enter the sums (the mock sees its NaN y? no: its y)
			if (tr) {
				S += (double)(r + 1) * d.y[i] * (weighted ? d.w[i] : 1.0);
				++n_train;
			}
		}
		if (counted < 2) {
			CHECK(out.is_null[k]);
			continue;
		}
		CHECK(!out.is_null[k] && (size_t)(out.offsets[k + 1] - out.offsets[k]) == rows.size());
		for (size_t r = 0; r < rows.size(); ++r) {
			const size_t i = rows[r], at = (size_t)out.offsets[k] + r;
			const uint8_t fl = out.flags[at];
			CHECK(((fl & 1) != 0) == (d.y_null[i] != 0));
			if (!d.y_null[i]) CHECK(out.vals[at * 4] == d.y[i]);
			CHECK(((fl & 16) != 0) == (train[r] != 0));
			const bool x_nan = d.xe_null[i * p] || d.xe_null[i * p + p - 1];
			if (x_nan) {
				CHECK((fl & 14) == 14); // a non-finite prediction: three NULLs
			} else {
				const double v = S + d.x[i * p] + d.x[i * p + p - 1];
				CHECK((fl & 14) == 0 && out.vals[at * 4 + 1] == v && out.vals[at * 4 + 2] == v - (double)n_train && out.vals[at * 4 + 3] == v + (double)counted);
			}
		}
	}
}

static void family_vif(size_t p, int threads, size_t vsize, bool dictionary) {
	const size_t n = 3000, K = 23;
	Data d = make_data(n, p, K, 31 + (unsigned)p, true);
	for (size_t i = 0; i < n; ++i)
		if (d.key[i] == 5 && i % 50 == 0) d.x[i * p + 1] = NAN; // key 5: column 1 comes out shorter -> NULL
	for (size_t i = 0; i < n; ++i)
		if (d.key[i] == 6) d.x_null[i] = i % 2 || i > 40;        // key 6: (almost) no rows
	FamilyQuery q("anofox_stats_vif_agg", nullptr, false, false);
	const int before = g_vif_calls.load();
	FamilyOut out = q.GroupBy(d.in(), nullptr, d.key.data(), K, threads, vsize, dictionary);
	CHECK(g_vif_calls.load() - before <= 1);
	for (size_t k = 0; k < K; ++k) {
		std::vector<size_t> rows;
		for (int t = 0; t < threads; ++t)
			for (size_t i = 0; i < n; ++i)
				if (d.key[i] == k && (int)((i / vsize) % (size_t)threads) == t && !d.x_null[i]) rows.push_back(i);
		bool ragged = false;
		for (size_t i : rows) ragged = ragged || isnan(d.x[i * p + 1]);
		if (p < 2 || rows.size() < 3 || ragged) {
			CHECK(out.is_null[k]);
			continue;
		}
		CHECK(!out.is_null[k] && (size_t)(out.offsets[k + 1] - out.offsets[k]) == p);
		for (size_t j = 0; j < p; ++j) {
			double s = 0.0;
			for (size_t r = 0; r < rows.size(); ++r) s += (double)(r + 1) * d.x[rows[r] * p + j];
			CHECK(out.vals[(size_t)out.offsets[k] + j] == s);
		}
	}
}

// the window aggregates: a frame per output row (naive aggregator) and leaves combined under PRESERVE_INPUT (segment tree)
static void family_window(const char *fn, size_t p, bool intercept) {
	const bool weighted = std::string(fn).find("wls") != std::string::npos;
	const size_t n = 600;
	Data d = make_data(n, p, 1, 17 + (unsigned)p, true);
	FamilyQuery q(fn, intercept ? nullptr : "fit_intercept=false", false, false);
	auto expect = [&](const std::vector<size_t> &frame, double *v3) -> bool { // false: NULL
		double S = 0.0;
		int64_t n_train = 0;
		bool has_current = false;
		size_t current = 0, pos = 0;
		for (size_t i : frame) {
			if (d.x_null[i]) {
				has_current = false;
				continue;
			}
			has_current = true;
			current = i;
			if (d.y_null[i] || (weighted && d.w_null[i])) continue;
			S += (double)(++pos) * d.y[i] * (weighted ? d.w[i] : 1.0);
			++n_train;
		}
		if (!has_current || n_train <= (int64_t)(p + (intercept ? 1 : 0))) return false;
		const double v = S + d.x[current * p] + d.x[current * p + p - 1];
		v3[0] = v;
		v3[1] = v - (double)n_train;
		v3[2] = v + (double)n_train;
		return true;
	};
	for (size_t preceding : {(size_t)11, (size_t)40}) {
		FamilyOut out = q.Window(d.in(), preceding, 128);
		CHECK(out.is_null.size() == n);
		for (size_t o = 0; o < n; ++o) {
			std::vector<size_t> frame;
			for (size_t r = o >= preceding ? o - preceding : 0; r <= o; ++r) frame.push_back(r);
			double v[3];
			const bool ok = expect(frame, v);
			CHECK(ok == !out.is_null[o]);
			if (ok) CHECK(out.vals[o * 3] == v[0] && out.vals[o * 3 + 1] == v[1] && out.vals[o * 3 + 2] == v[2]);
		}
	}
	const size_t leaf = 7, back = 4;
	FamilyOut tree = q.TreeWindow(d.in(), leaf, back, 64);
	const size_t n_leaves = (n + leaf - 1) / leaf;
	CHECK(tree.is_null.size() == n_leaves);
	for (size_t o = 0; o < n_leaves; ++o) {
		std::vector<size_t> frame;
		for (size_t l = o >= back ? o - back : 0; l <= o; ++l)
			for (size_t r = l * leaf; r < std::min(n, (l + 1) * leaf); ++r) frame.push_back(r);
		// Combine keeps the target's current row unless the source has one: replay leaf by leaf
		double S = 0.0;
		int64_t n_train = 0;
		bool has_current = false, initialised = false;
		size_t current = 0, pos = 0;
		for (size_t l = o >= back ? o - back : 0; l <= o; ++l) {
			bool leaf_init = false, leaf_has = false;
			size_t leaf_cur = 0;
			for (size_t r = l * leaf; r < std::min(n, (l + 1) * leaf); ++r) {
				if (d.x_null[r]) {
					if (leaf_init) leaf_has = false;
					continue;
				}
				leaf_init = true;
				leaf_has = true;
				leaf_cur = r;
				if (d.y_null[r] || (weighted && d.w_null[r])) continue;
				S += (double)(++pos) * d.y[r] * (weighted ? d.w[r] : 1.0);
				++n_train;
			}
			if (!leaf_init) continue; // an uninitialised source is skipped
			if (!initialised) {
				initialised = true;
				has_current = leaf_has;
				current = leaf_cur;
			} else if (leaf_has) {
				has_current = true;
				current = leaf_cur;
			}
		}
		const bool ok = initialised && has_current && n_train > (int64_t)(p + (intercept ? 1 : 0));
		CHECK(ok == !tree.is_null[o]);
		if (ok) {
			const double v = S + d.x[current * p] + d.x[current * p + p - 1];
			CHECK(tree.vals[o * 3] == v && tree.vals[o * 3 + 1] == v - (double)n_train && tree.vals[o * 3 + 2] == v + (double)n_train);
		}
	}
}

static void family_errors() {
	// a group whose rows disagree in width
	const size_t n = 10, p = 3;
	Data d = make_data(n, p, 1, 5, false);
	std::vector<uint32_t> len(n, 3);
	len[6] = 2;
	Inputs in = d.in();
	in.x_len = len.data();
	for (const char *fn : {"ols_fit_predict_agg", "ridge_fit_predict_agg", "vif_agg"}) {
		std::string msg;
		try {
			FamilyQuery q(fn, nullptr, false, false);
			q.GroupBy(in, nullptr, d.key.data(), 1, 1, 2048, false);
		} catch (const std::exception &e) {
			msg = e.what();
		}
		CHECK(msg.find("Inconsistent feature count") != std::string::npos);
		if (std::string(fn) != "ridge_fit_predict_agg") CHECK(msg.find("expected 3, got 2") != std::string::npos);
	}
}

int main() {
	registration_and_bind();
	group_by("anofox_stats_ols_fit_agg", 3, 1, 2048, false);
	group_by("ols_fit_agg", 3, 4, 64, true);
	group_by("anofox_stats_wls_fit_agg", 8, 3, 100, true);
	group_by("ridge_fit_agg", 12, 5, 2048, false);
	// (r4) the query's states hash-partitioned over three device states: thread-local sources meet targets on other shards in
	// Combine (records exported / imported / merged), every key's rows still arrive once and in order
	setenv("ANOFOX_HIP_DEVICES", "0,0,0", 1);
	{
		const int before = g_export_calls.load();
		group_by("anofox_stats_ols_fit_agg", 3, 4, 64, false, 3);
		group_by("anofox_stats_wls_fit_agg", 8, 3, 100, true, 3);
		CHECK(g_export_calls.load() > before && g_import_calls.load() > 0); // some pairs did cross shards
		group_by("ridge_fit_agg", 12, 5, 2048, false, 1);                   // a log-only width stays on one shard
	}
	unsetenv("ANOFOX_HIP_DEVICES");
	prepared_twice();
	window_replay();
	errors_and_flags();
	family_registration();
	family_group_by("anofox_stats_ols_fit_predict_agg", 3, 1, 2048, false, false, nullptr);
	family_group_by("ols_predict_agg", 4, 4, 64, true, true, nullptr);
	family_group_by("ridge_fit_predict_agg", 2, 3, 100, true, false, "null_policy=drop_y_zero_x;alpha=2");
	family_group_by("anofox_stats_wls_fit_predict_agg", 5, 2, 333, false, true, "null_policy=drop_y_zero_x");
	family_group_by("wls_predict_agg", 1, 5, 50, true, false, nullptr);
	family_vif(4, 3, 128, true);
	family_vif(2, 1, 2048, false);
	family_window("anofox_stats_ols_fit_predict", 2, true);
	family_window("ridge_fit_predict", 3, false);
	family_window("wls_fit_predict", 1, true);
	family_errors();
	CHECK(g_contexts == 0 && g_states == 0 && g_host_allocs == 0); // everything released
	printf("glue_sanitize: all scenarios passed\n");
	return 0;
}
