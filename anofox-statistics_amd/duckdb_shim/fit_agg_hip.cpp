// fit_agg_hip.cpp — DuckDB-side shim for the three aggregates.
//
// Drop-in replacements for the Finalize callbacks of the reference's
//   src/aggregate_functions/ols_aggregate.cpp:249-338    (OlsAggFinalize)
//   src/aggregate_functions/ridge_aggregate.cpp:255-345  (RidgeAggFinalize)
//   src/aggregate_functions/wls_aggregate.cpp:268-362    (WlsAggFinalize)
// which loop one anofox_*_fit FFI call per group state.  Here the states of one Finalize vector (up to
// STANDARD_VECTOR_SIZE = 2048 groups) are packed into grouped columns and fitted with ONE call of
// anofox_hip_fit_batch_host (include/anofox_stats_hip.h).  Everything else of those files — state structs, Bind,
// Update, Combine, Destroy, registration under anofox_stats_*_fit_agg and the short aliases — stays as it is.
//
// NOT COMPILED IN THIS REPOSITORY: the reference's `duckdb` submodule (headers) is not available here; written
// against the DuckDB v1.4.5 / v1.5.5 API exactly as the reference uses it.  The three
// wrapper functions at the bottom have the signature DuckDB expects for `aggregate_finalize_t`.
#include <vector>

#include "duckdb.hpp"
#include "duckdb/function/aggregate_function.hpp"

#include "../include/anofox_stats_ffi.h" // the reference's header: structs and enums
#include "../include/ffi_enum_converters.hpp"
#include "anofox_stats_hip.h"            // after the reference's header: adds only the batch API

namespace duckdb {

namespace {

// Field access differs per model only in the weights buffer and the ridge options.
struct OlsTraits {
	static constexpr AnofoxHipModel kModel = ANOFOX_HIP_MODEL_OLS;
	template <class STATE> static const vector<double> *Weights(const STATE &) { return nullptr; }
	template <class STATE> static void Fill(const STATE &s, AnofoxHipBatchOptions &o) { o.hc_type = ConvertHcType(s.hc_type); }
};
struct RidgeTraits {
	static constexpr AnofoxHipModel kModel = ANOFOX_HIP_MODEL_RIDGE;
	template <class STATE> static const vector<double> *Weights(const STATE &) { return nullptr; }
	template <class STATE> static void Fill(const STATE &s, AnofoxHipBatchOptions &o) {
		o.alpha = s.alpha;
		o.lambda_scaling = ConvertLambdaScaling(s.lambda_scaling);
		o.hc_type = ANOFOX_HC_NONE;
	}
};
struct WlsTraits {
	static constexpr AnofoxHipModel kModel = ANOFOX_HIP_MODEL_WLS;
	template <class STATE> static const vector<double> *Weights(const STATE &s) { return &s.weights; }
	template <class STATE> static void Fill(const STATE &s, AnofoxHipBatchOptions &o) { o.hc_type = ConvertHcType(s.hc_type); }
};

void AppendList(Vector &list_vec, idx_t row, const double *src, idx_t n) {
	auto entries = ListVector::GetData(list_vec);
	auto offset = ListVector::GetListSize(list_vec);
	ListVector::Reserve(list_vec, offset + n); // the reference's SetListInResult omits this (ols_aggregate.cpp:237-246)
	auto child = FlatVector::GetData<double>(ListVector::GetEntry(list_vec));
	for (idx_t k = 0; k < n; k++) child[offset + k] = src[k];
	entries[row].offset = offset;
	entries[row].length = n;
	ListVector::SetListSize(list_vec, offset + n);
}

template <class STATE, class TRAITS>
void BatchedFinalize(Vector &state_vector, Vector &result, idx_t count, idx_t offset) {
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (STATE **)sdata.data;

	// 1. grouped columns: rows of state i are [offsets[i], offsets[i+1]); uninitialised states hold no rows and
	//    come back with status 100 (the "< 2 rows -> NULL" rule of ols_aggregate.cpp:263-267)
	vector<int64_t> offsets(count + 1, 0);
	idx_t p = 0;
	const STATE *any = nullptr;
	for (idx_t i = 0; i < count; i++) {
		auto &st = *states[sdata.sel->get_index(i)];
		offsets[i + 1] = offsets[i] + (st.initialized ? (int64_t)st.y_values.size() : 0);
		if (st.initialized) {
			p = st.n_features;
			any = &st;
		}
	}
	if (!any) {
		for (idx_t i = 0; i < count; i++) FlatVector::SetNull(result, i + offset, true);
		return;
	}
	const idx_t n_rows = (idx_t)offsets[count];
	vector<double> y(n_rows), w;
	vector<vector<double>> x(p, vector<double>(n_rows));
	const bool weighted = TRAITS::Weights(*any) != nullptr;
	if (weighted) w.resize(n_rows);
	for (idx_t i = 0; i < count; i++) {
		auto &st = *states[sdata.sel->get_index(i)];
		if (!st.initialized) continue;
		std::copy(st.y_values.begin(), st.y_values.end(), y.begin() + offsets[i]);
		for (idx_t j = 0; j < p; j++) std::copy(st.x_columns[j].begin(), st.x_columns[j].end(), x[j].begin() + offsets[i]);
		if (weighted) {
			auto *sw = TRAITS::Weights(st);
			std::copy(sw->begin(), sw->end(), w.begin() + offsets[i]);
		}
	}
	vector<const double *> x_cols(p);
	for (idx_t j = 0; j < p; j++) x_cols[j] = x[j].data();

	// 2. one GPU call for the whole vector of groups (options are per query: every state carries the bind data)
	AnofoxHipBatchOptions opt {};
	opt.model = TRAITS::kModel;
	opt.fit_intercept = any->fit_intercept;
	opt.compute_inference = any->compute_inference;
	opt.confidence_level = any->confidence_level;
	opt.solver = ConvertSolverType(any->solver);
	opt.alpha = 1.0;
	TRAITS::Fill(*any, opt);
	vector<double> core(count * (p + 6)), inf(opt.compute_inference ? count * (5 * p + 2) : 0);
	AnofoxError err;
	if (!anofox_hip_fit_batch_host(/*per-thread default context*/ nullptr, (int64_t)count, p, (int64_t)n_rows,
	                               offsets.data(), y.data(), x_cols.data(), weighted ? w.data() : nullptr, opt,
	                               core.data(), inf.empty() ? nullptr : inf.data(), &err)) {
		throw InvalidInputException("anofox_stats fit_agg (HIP): %s", err.message);
	}

	// 3. records -> STRUCT (field order of GetOlsAggResultType, ols_aggregate.cpp:74-96)
	auto &entries = StructVector::GetEntries(result);
	for (idx_t i = 0; i < count; i++) {
		const double *rec = &core[i * (p + 6)];
		const idx_t r = i + offset;
		auto &st = *states[sdata.sel->get_index(i)];
		if (rec[p + 5] != 0.0) { // NULL group: too few rows or a failed fit (ols_aggregate.cpp:263-267,298-301)
			FlatVector::SetNull(result, r, true);
			st.Reset();
			continue;
		}
		AppendList(*entries[0], r, rec, p);
		FlatVector::GetData<double>(*entries[1])[r] = rec[p];
		FlatVector::GetData<double>(*entries[2])[r] = rec[p + 1];
		FlatVector::GetData<double>(*entries[3])[r] = rec[p + 2];
		FlatVector::GetData<double>(*entries[4])[r] = rec[p + 3];
		FlatVector::GetData<int64_t>(*entries[5])[r] = (int64_t)rec[p + 4];
		FlatVector::GetData<int64_t>(*entries[6])[r] = (int64_t)p;
		if (opt.compute_inference) {
			const double *ir = &inf[i * (5 * p + 2)];
			for (idx_t k = 0; k < 5; k++) AppendList(*entries[7 + k], r, ir + k * p, p); // se, t, p, ci_lower, ci_upper
			FlatVector::GetData<double>(*entries[12])[r] = ir[5 * p];
			FlatVector::GetData<double>(*entries[13])[r] = ir[5 * p + 1];
		}
		st.Reset();
	}
}

} // namespace

// aggregate_finalize_t wrappers: plug these into the AggregateFunction constructors at
// ols_aggregate.cpp:381-386, ridge_aggregate.cpp:392-397, wls_aggregate.cpp:405-410.
void OlsAggFinalizeHip(Vector &state_vector, AggregateInputData &, Vector &result, idx_t count, idx_t offset) {
	BatchedFinalize<OlsAggregateState, OlsTraits>(state_vector, result, count, offset);
}
void RidgeAggFinalizeHip(Vector &state_vector, AggregateInputData &, Vector &result, idx_t count, idx_t offset) {
	BatchedFinalize<RidgeAggregateState, RidgeTraits>(state_vector, result, count, offset);
}
void WlsAggFinalizeHip(Vector &state_vector, AggregateInputData &, Vector &result, idx_t count, idx_t offset) {
	BatchedFinalize<WlsAggregateState, WlsTraits>(state_vector, result, count, offset);
}

} // namespace duckdb
