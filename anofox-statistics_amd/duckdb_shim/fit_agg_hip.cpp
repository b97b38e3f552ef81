// fit_agg_hip.cpp — DuckDB glue of the three aggregates over the GPU-resident state.
//
// Replaces, in the reference's
//   src/aggregate_functions/ols_aggregate.cpp    (state :19-42, Initialize :103, Destroy :108, Update :120-186,
//                                                 Combine :189-234, Finalize :249-338)
//   src/aggregate_functions/ridge_aggregate.cpp  (:19-43, :124-191, :194-240, :255-345)
//   src/aggregate_functions/wls_aggregate.cpp    (:19-44, :122-201, :204-253, :268-362)
// the per-group std::vector row buffers and the one-FFI-call-per-group Finalize.  The DuckDB state shrinks to a
// slot number; the rows go through anofox_shim::AggArena (agg_arena.hpp: page-locked chunk buffers ->
// anofox_hip_agg_state_update_host) into one O(p^2) moment record per slot on the GPU; Finalize reads the records of
// one batched solve.  Bind, the result type, the options parser and the registration (names, aliases, overloads:
// ols_aggregate.cpp:74-96,343-426) stay exactly as they are — only the five callbacks and the state size change:
//
//   AggregateFunction(name, args, LogicalType::ANY, AggregateFunction::StateSize<HipAggState>, HipAggInitialize,
//                     HipAggUpdate<OlsTraits>, HipAggCombine, HipAggFinalize<OlsTraits>, nullptr, OlsAggBind, HipAggDestroy)
//
// and the bind data gains one member, `shared_ptr<anofox_shim::AggArena> arena`, created in Bind from the parsed
// options and shared by Copy() (every thread of the query must reach the same state).
//
// NOT COMPILED IN THIS REPOSITORY: the reference's `duckdb` submodule (headers) is empty here; written against the
// DuckDB v1.4.5 / v1.5.5 API exactly as the reference uses it.  All logic that does not need
// DuckDB types lives in agg_arena.hpp, which IS compiled and tested here (arena_capi.cpp, tests/test_gpu_arena.py).
#include "duckdb.hpp"
#include "duckdb/function/aggregate_function.hpp"

#include "../include/anofox_stats_ffi.h" // the reference's header: structs and enums
#include "../include/ffi_enum_converters.hpp"
#include "anofox_stats_hip.h"            // after the reference's header: adds only the batch / state API
#include "agg_arena.hpp"

namespace duckdb {

// The whole DuckDB-side aggregate state: which slot of the query's GPU state this group (of this thread's hash
// table) owns.  -1 until the first Update touches it (Initialize has no access to the bind data).
struct HipAggState {
	int64_t slot;
};

// What the three bind-data classes add (OlsAggregateBindData etc. keep their option fields):
struct HipAggBindMixin {
	shared_ptr<anofox_shim::AggArena> arena;
};

struct OlsTraits {
	static constexpr bool kWeighted = false;
	using BindData = OlsAggregateBindData; // + HipAggBindMixin
};
struct RidgeTraits {
	static constexpr bool kWeighted = false;
	using BindData = RidgeAggregateBindData;
};
struct WlsTraits {
	static constexpr bool kWeighted = true;
	using BindData = WlsAggregateBindData;
};

// Bind-time helper: the batch options of the query from the parsed bind data (called at the end of *AggBind).
template <class BIND>
AnofoxHipBatchOptions MakeHipOptions(const BIND &b, AnofoxHipModel model) {
	AnofoxHipBatchOptions o {};
	o.model = model;
	o.fit_intercept = b.fit_intercept;
	o.compute_inference = b.compute_inference;
	o.confidence_level = b.confidence_level;
	o.solver = ConvertSolverType(b.solver);
	o.alpha = 1.0;
	return o; // ridge: o.alpha = b.alpha, o.lambda_scaling = ConvertLambdaScaling(b.lambda_scaling); ols/wls: o.hc_type
}

static void HipAggInitialize(const AggregateFunction &, data_ptr_t state_p) {
	reinterpret_cast<HipAggState *>(state_p)->slot = -1;
}

static void HipAggDestroy(Vector &, AggregateInputData &, idx_t) {
	// nothing per state: the slots belong to the arena, which the bind data's shared_ptr releases with the query
}

// Update: ols_aggregate.cpp:120-186 / wls_aggregate.cpp:122-201 with the push_backs replaced by one arena append.
template <class TRAITS>
static void HipAggUpdate(Vector inputs[], AggregateInputData &aggr_input_data, idx_t input_count, Vector &state_vector, idx_t count) {
	auto &bind = aggr_input_data.bind_data->Cast<typename TRAITS::BindData>();
	auto &arena = *bind.arena;
	UnifiedVectorFormat y_data, x_data, w_data, sdata;
	inputs[0].ToUnifiedFormat(count, y_data);
	inputs[1].ToUnifiedFormat(count, x_data);
	if (TRAITS::kWeighted) inputs[2].ToUnifiedFormat(count, w_data);
	auto y_values = UnifiedVectorFormat::GetData<double>(y_data);
	auto w_values = TRAITS::kWeighted ? UnifiedVectorFormat::GetData<double>(w_data) : nullptr;
	auto x_list = ListVector::GetData(inputs[1]);
	auto &x_child = ListVector::GetEntry(inputs[1]);
	auto x_child_data = FlatVector::GetData<double>(x_child);
	auto &x_child_validity = FlatVector::Validity(x_child);
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipAggState **)sdata.data;

	anofox_shim::AggArena::Writer writer(arena); // one lock per vector
	double row[128]; // anofox_hip_max_features(); everything goes to the GPU state (moments up to 8 features, the rows themselves beyond)
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		if (state.slot < 0) state.slot = writer.NewSlot(); // the group exists even if every row of it is skipped
		auto y_idx = y_data.sel->get_index(i);
		if (!y_data.validity.RowIsValid(y_idx)) continue;                       // ols_aggregate.cpp:150-153
		auto x_idx = x_data.sel->get_index(i);
		if (!x_data.validity.RowIsValid(x_idx)) continue;                       // :157-159
		double w = 1.0;
		if (TRAITS::kWeighted) {
			auto w_idx = w_data.sel->get_index(i);
			if (!w_data.validity.RowIsValid(w_idx)) continue;                   // wls_aggregate.cpp:160-166
			w = w_values[w_idx];
		}
		auto entry = x_list[x_idx];
		if (entry.length > 128) throw InvalidInputException("anofox_stats fit_agg (HIP): at most 128 features are supported");
		const idx_t n = entry.length;
		for (idx_t j = 0; j < n; j++) // a NULL list element becomes NaN: the fit's row filter drops the row (ols.rs:59-66)
			row[j] = x_child_validity.RowIsValid(entry.offset + j) ? x_child_data[entry.offset + j] : NAN;
		try {
			writer.Append((uint32_t)state.slot, y_values[y_idx], row, entry.length, w);
		} catch (const std::invalid_argument &e) {
			throw InvalidInputException(e.what());                              // "Inconsistent feature count: ..." (:172-175)
		}
	}
}

// Combine: ols_aggregate.cpp:189-234.  A source without a slot has seen no Update; a target without one adopts the
// source's slot (the reference moves the buffers); otherwise the pair is merged on the GPU.
static void HipAggCombine(Vector &source_vector, Vector &target_vector, AggregateInputData &aggr_input_data, idx_t count) {
	UnifiedVectorFormat source_data, target_data;
	source_vector.ToUnifiedFormat(count, source_data);
	target_vector.ToUnifiedFormat(count, target_data);
	auto sources = (HipAggState **)source_data.data;
	auto targets = (HipAggState **)target_data.data;
	vector<uint32_t> src, dst;
	for (idx_t i = 0; i < count; i++) {
		auto &source = *sources[source_data.sel->get_index(i)];
		auto &target = *targets[target_data.sel->get_index(i)];
		if (source.slot < 0) continue;
		if (target.slot < 0) {
			target.slot = source.slot;
			source.slot = -1;
			continue;
		}
		src.push_back((uint32_t)source.slot);
		dst.push_back((uint32_t)target.slot);
	}
	if (src.empty()) return;
	auto &arena = *aggr_input_data.bind_data->Cast<HipAggBindMixin>().arena;
	arena.Combine(src.data(), dst.data(), src.size());
}

static void AppendList(Vector &list_vec, idx_t row, const double *src, idx_t n) {
	auto entries = ListVector::GetData(list_vec);
	auto offset = ListVector::GetListSize(list_vec);
	ListVector::Reserve(list_vec, offset + n); // the reference's SetListInResult omits this (ols_aggregate.cpp:237-246)
	auto child = FlatVector::GetData<double>(ListVector::GetEntry(list_vec));
	for (idx_t k = 0; k < n; k++) child[offset + k] = src[k];
	entries[row].offset = offset;
	entries[row].length = n;
	ListVector::SetListSize(list_vec, offset + n);
}

// Finalize: ols_aggregate.cpp:249-338.  The first call of the query solves every slot at once; each call then only
// copies its <= 2048 records into the STRUCT vector (field order of GetOlsAggResultType, :74-96).
template <class TRAITS>
static void HipAggFinalize(Vector &state_vector, AggregateInputData &aggr_input_data, Vector &result, idx_t count, idx_t offset) {
	auto &bind = aggr_input_data.bind_data->Cast<typename TRAITS::BindData>();
	auto &arena = *bind.arena;
	arena.Solve();
	const idx_t p = arena.FeatureCount();
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipAggState **)sdata.data;
	auto &entries = StructVector::GetEntries(result);
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		const idx_t r = i + offset;
		const double *rec = state.slot < 0 ? nullptr : arena.Core((uint32_t)state.slot);
		if (!rec) { // fewer than 2 accumulated rows or a failed fit -> NULL, the query continues (:263-267,298-301)
			FlatVector::SetNull(result, r, true);
			continue;
		}
		AppendList(*entries[0], r, rec, p);
		FlatVector::GetData<double>(*entries[1])[r] = rec[p];
		FlatVector::GetData<double>(*entries[2])[r] = rec[p + 1];
		FlatVector::GetData<double>(*entries[3])[r] = rec[p + 2];
		FlatVector::GetData<double>(*entries[4])[r] = rec[p + 3];
		FlatVector::GetData<int64_t>(*entries[5])[r] = (int64_t)rec[p + 4];
		FlatVector::GetData<int64_t>(*entries[6])[r] = (int64_t)p;
		if (bind.compute_inference) {
			const double *ir = arena.Inference((uint32_t)state.slot);
			for (idx_t k = 0; k < 5; k++) AppendList(*entries[7 + k], r, ir + k * p, p); // se, t, p, ci_lower, ci_upper
			FlatVector::GetData<double>(*entries[12])[r] = ir[5 * p];
			FlatVector::GetData<double>(*entries[13])[r] = ir[5 * p + 1];
		}
	}
}

} // namespace duckdb
