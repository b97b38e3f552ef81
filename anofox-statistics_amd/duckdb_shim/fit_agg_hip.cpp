// fit_agg_hip.cpp — DuckDB glue of {ols,ridge,wls}_fit_agg over the GPU-resident aggregate state.
//
// Replaces, in the reference's
//   src/aggregate_functions/ols_aggregate.cpp    (state :19-42, bind data :47-69, result type :74-96, Initialize :103,
//                                                 Destroy :108-118, Update :120-186, Combine :189-234, Finalize :249-338,
//                                                 Bind :343-372, registration :377-426)
//   src/aggregate_functions/ridge_aggregate.cpp  (:19-43, :49-76, :78-104, :124-191, :194-240, :255-345, :350-385, :388-440)
//   src/aggregate_functions/wls_aggregate.cpp    (:19-44, :49-74, :76-102, :122-201, :204-253, :268-362, :367-398, :401-452)
// the per-group std::vector row buffers and the one-FFI-call-per-group Finalize: the extension's entry point calls
// RegisterHip{Ols,Ridge,Wls}AggregateFunction(loader) instead of Register{Ols,Ridge,Wls}AggregateFunction(loader) and
// the SQL surface stays what it was — names anofox_stats_*_fit_agg with the aliases *_fit_agg, the overloads with and
// without the constant options argument, the 7 / 14-field STRUCT result, NULL for groups that cannot be fitted.
//
// The DuckDB state shrinks to a slot number.  Rows go through anofox_shim::AggArena (agg_arena.hpp: lock-free appends
// into the calling thread's page-locked chunk -> anofox_hip_agg_state_update_host) into one O(p^2) moment record per
// slot on the GPU (or the logged rows, for designs wider than 8 features and HC errors); Finalize asks the arena for
// the records of its vector of states, which fits whatever changed since it was last fitted in ONE batched call.
// The bind data owns the arena and Copy() shares it: every thread of the query reaches the same device state.
//
// Compiled and driven in this repository against a stand-in of DuckDB's headers (tests/tools/duckdb_stub, -Wall -Wextra
// under ASan / UBSan with a mock of the C ABI; on the GPU with the real library: tests/test_gpu_glue.py).  Written
// against the DuckDB v1.4.5 / v1.5.5 API exactly as the reference uses it.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <cctype>
#include <map>
#include <memory>
#include <mutex>

#include "duckdb.hpp"
#include "duckdb/common/types/data_chunk.hpp"
#include "duckdb/execution/expression_executor.hpp"
#include "duckdb/function/aggregate_function.hpp"
#include "duckdb/main/extension/extension_loader.hpp"
#include "duckdb/parser/parsed_data/create_aggregate_function_info.hpp"

#include "anofox_stats_hip.h" // coexists with the reference's anofox_stats_ffi.h (same structs and enums, guarded)
#include "agg_arena.hpp"
#include "sharded_arena.hpp" // hash64: the routing hash of SURVEY.md 8(e)
#include "fit_agg_hip.hpp"
#include "hip_options.hpp"

#include <thread>

namespace duckdb {

namespace {

// The whole DuckDB-side aggregate state: the group's feature count, fixed by its first accepted row exactly as in the
// reference (ols_aggregate.cpp:164-175: per STATE, so groups of one query may differ in width), and which slot of the
// query's GPU state of that width the group (of this thread's hash table) owns.
struct HipAggState {
	int64_t slot;       // -1 until the first accepted row (Initialize has no access to the bind data)
	int64_t n_features; // -1 = no accepted row yet
	int64_t shard;      // (r4) which of the query's device states holds the slot: hash64(state address) % W at the first accepted row
};

using namespace hip_glue;

// ---- the query's device states: per feature count that occurs (normally one), W AggArenas — one per GPU of the node ----
// (r4, SURVEY.md 8e / north_star: "groups are hash-partitioned across the GPUs of one node").  DuckDB's callbacks see state
// pointers, not keys, so a state is routed at its first accepted row by hash64(state address) % W and stays there; a Combine
// whose source and target sit on different devices moves the source's O(p^2) moment record (528 bytes at p = 8) through the
// host — anofox_hip_agg_state_export_slots_host / import — and merges it on the target's device; Finalize fits every device's
// states in one batched call per device, concurrently.  ANOFOX_HIP_DEVICES: a count ("8" = devices 0 .. 7) or a list of
// ordinals ("0,1,2,3"; "0,0" = two shards on one GPU, the single-GPU rehearsal of the tests); unset = one shard on the
// calling thread's current device.  Designs of more than 8 features and HC standard errors keep ROWS on the device, not
// moment records (log-only states): those stay on shard 0.
struct HipArenaSet {
	explicit HipArenaSet(const AnofoxHipBatchOptions &o) : options(o) {
		if (const char *v = getenv("ANOFOX_HIP_DEVICES")) {
			string str(v);
			if (str.find(',') == string::npos) {
				const int n = atoi(str.c_str());
				for (int k = 0; k < n && k < 64; ++k) devices.push_back(k);
			} else {
				size_t pos = 0;
				while (pos <= str.size() && devices.size() < 64) {
					const size_t next = str.find(',', pos);
					const string item = str.substr(pos, next == string::npos ? string::npos : next - pos);
					if (!item.empty()) devices.push_back(atoi(item.c_str()));
					if (next == string::npos) break;
					pos = next + 1;
				}
			}
		}
		if (devices.empty()) devices.push_back(-1);
	}
	// shards a state of this width may live on
	uint32_t ShardCount(size_t p) const {
		const bool log_only = p > 8 || (options.compute_inference && options.hc_type != ANOFOX_HC_NONE && options.model != ANOFOX_HIP_MODEL_RIDGE);
		return log_only ? 1u : (uint32_t)devices.size();
	}
	uint32_t ShardOf(const void *state, size_t p) const {
		const uint32_t w = ShardCount(p);
		return w <= 1 ? 0u : (uint32_t)(anofox_shim::hash64((uint64_t)(uintptr_t)state) % w);
	}
	anofox_shim::AggArena &For(size_t p, uint32_t shard) {
		std::lock_guard<std::mutex> lk(mu);
		auto &slot = arenas[std::make_pair(p, shard)];
		if (!slot) {
			// (the row-log budgets are per query: split over the shards)
			const size_t w = ShardCount(p);
			slot.reset(new anofox_shim::AggArena(options, (size_t)1 << 18, ((size_t)64 << 30) / w, ((size_t)32 << 30) / w, devices[shard % devices.size()]));
		}
		return *slot;
	}
	template <class F>
	void ForEach(F &&f) {
		std::lock_guard<std::mutex> lk(mu);
		for (auto &kv : arenas) f(kv.first.first, *kv.second);
	}
	AnofoxHipBatchOptions options;
	vector<int> devices;
	std::mutex mu;
	std::map<std::pair<size_t, uint32_t>, std::unique_ptr<anofox_shim::AggArena>> arenas;
};

// ---- bind data: the parsed options and the query's arenas; Copy() shares them ----
struct HipAggBindData : public FunctionData {
	HipAggBindData(HipModel model_p, const HipFitOptions &opts_p)
	    : model(model_p), opts(opts_p), arenas(make_shared_ptr<HipArenaSet>(MakeHipOptions(model_p, opts_p))) {}
	HipAggBindData(HipModel model_p, const HipFitOptions &opts_p, shared_ptr<HipArenaSet> arenas_p)
	    : model(model_p), opts(opts_p), arenas(std::move(arenas_p)) {}
	HipModel model;
	HipFitOptions opts;
	shared_ptr<HipArenaSet> arenas;

	unique_ptr<FunctionData> Copy() const override { return make_uniq<HipAggBindData>(model, opts, arenas); }
	bool Equals(const FunctionData &other_p) const override {
		auto &other = other_p.Cast<HipAggBindData>();
		return model == other.model && opts == other.opts && arenas == other.arenas;
	}
};

// result type: GetOlsAggResultType (ols_aggregate.cpp:74-96) and its ridge / wls twins
LogicalType GetHipAggResultType(bool compute_inference) {
	child_list_t<LogicalType> children;
	children.push_back(make_pair("coefficients", LogicalType::LIST(LogicalType::DOUBLE)));
	children.push_back(make_pair("intercept", LogicalType::DOUBLE));
	children.push_back(make_pair("r_squared", LogicalType::DOUBLE));
	children.push_back(make_pair("adj_r_squared", LogicalType::DOUBLE));
	children.push_back(make_pair("residual_std_error", LogicalType::DOUBLE));
	children.push_back(make_pair("n_observations", LogicalType::BIGINT));
	children.push_back(make_pair("n_features", LogicalType::BIGINT));
	if (compute_inference) {
		children.push_back(make_pair("std_errors", LogicalType::LIST(LogicalType::DOUBLE)));
		children.push_back(make_pair("t_values", LogicalType::LIST(LogicalType::DOUBLE)));
		children.push_back(make_pair("p_values", LogicalType::LIST(LogicalType::DOUBLE)));
		children.push_back(make_pair("ci_lower", LogicalType::LIST(LogicalType::DOUBLE)));
		children.push_back(make_pair("ci_upper", LogicalType::LIST(LogicalType::DOUBLE)));
		children.push_back(make_pair("f_statistic", LogicalType::DOUBLE));
		children.push_back(make_pair("f_pvalue", LogicalType::DOUBLE));
	}
	return LogicalType::STRUCT(std::move(children));
}

template <HipModel MODEL>
unique_ptr<FunctionData> HipAggBind(ClientContext &context, AggregateFunction &function, vector<unique_ptr<Expression>> &arguments) {
	HipFitOptions opts;
	// the optional last argument: y, x[, weight][, options] — parsed only when it folds to a constant, as upstream
	// (ols_aggregate.cpp:348, ridge_aggregate.cpp:355, wls_aggregate.cpp:372)
	const idx_t opt_idx = MODEL == HipModel::WLS ? 3 : 2;
	if (arguments.size() > opt_idx && arguments[opt_idx]->IsFoldable()) ParseHipFitOptions(ExpressionExecutor::EvaluateScalar(context, *arguments[opt_idx]), opts);
	function.return_type = GetHipAggResultType(opts.compute_inference);
	return make_uniq<HipAggBindData>(MODEL, opts);
}

void HipAggInitialize(const AggregateFunction &, data_ptr_t state_p) {
	auto &st = *reinterpret_cast<HipAggState *>(state_p);
	st.slot = -1;
	st.n_features = -1;
	st.shard = -1;
}

// Destroy: the state's slot goes back to its arena, which empties it on the device before handing it out again
// (ols_aggregate.cpp:108-118 runs the row buffers' destructors)
void HipAggDestroy(Vector &state_vector, AggregateInputData &aggr_input_data, idx_t count) {
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipAggState **)sdata.data;
	auto &arenas = *aggr_input_data.bind_data->Cast<HipAggBindData>().arenas;
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		if (state.slot >= 0) arenas.For((size_t)state.n_features, (uint32_t)state.shard).ReleaseSlot((uint32_t)state.slot);
		state.slot = -1;
		state.n_features = -1;
		state.shard = -1;
	}
}

// Update: ols_aggregate.cpp:120-186 / ridge_aggregate.cpp:124-191 / wls_aggregate.cpp:122-201 with the push_backs replaced
// by one append into the calling thread's chunk; the row's LIST entries are copied straight into page-locked memory.
template <HipModel MODEL>
void HipAggUpdate(Vector inputs[], AggregateInputData &aggr_input_data, idx_t input_count, Vector &state_vector, idx_t count) {
	constexpr bool kWeighted = MODEL == HipModel::WLS;
	if (input_count < (kWeighted ? 3u : 2u)) throw InvalidInputException("anofox_stats fit_agg (HIP): too few arguments");
	auto &arenas = *aggr_input_data.bind_data->Cast<HipAggBindData>().arenas;
	UnifiedVectorFormat y_data, x_data, w_data, sdata;
	inputs[0].ToUnifiedFormat(count, y_data);
	inputs[1].ToUnifiedFormat(count, x_data);
	if (kWeighted) inputs[2].ToUnifiedFormat(count, w_data);
	auto y_values = UnifiedVectorFormat::GetData<double>(y_data);
	auto w_values = kWeighted ? UnifiedVectorFormat::GetData<double>(w_data) : nullptr;
	auto x_list = UnifiedVectorFormat::GetData<list_entry_t>(x_data);
	auto &x_child = ListVector::GetEntry(inputs[1]);
	auto x_child_data = FlatVector::GetData<double>(x_child);
	auto &x_child_validity = FlatVector::Validity(x_child);
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipAggState **)sdata.data;
	const idx_t max_features = anofox_hip_max_features();

	// one Writer per (feature count, device shard) seen in this vector
	size_t cur_p = 0;
	uint32_t cur_shard = 0;
	anofox_shim::AggArena *cur_arena = nullptr;
	std::map<std::pair<size_t, uint32_t>, std::unique_ptr<anofox_shim::AggArena::Writer>> writers;
	anofox_shim::AggArena::Writer *writer = nullptr;
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		auto y_idx = y_data.sel->get_index(i);
		if (!y_data.validity.RowIsValid(y_idx)) continue; // ols_aggregate.cpp:150-153
		auto x_idx = x_data.sel->get_index(i);
		if (!x_data.validity.RowIsValid(x_idx)) continue; // :157-159
		double w = 1.0;
		if (kWeighted) {
			auto w_idx = w_data.sel->get_index(i);
			if (!w_data.validity.RowIsValid(w_idx)) continue; // wls_aggregate.cpp:160-166
			w = w_values[w_idx];
		}
		const auto entry = x_list[x_idx];
		// the first accepted row of a state fixes its feature count; every later one must agree (:164-175)
		if (state.n_features < 0) state.n_features = (int64_t)entry.length;
		if ((int64_t)entry.length != state.n_features)
			throw InvalidInputException("Inconsistent feature count: expected %llu, got %llu", (unsigned long long)state.n_features,
			                            (unsigned long long)entry.length);
		if (entry.length == 0) continue; // an empty x list: the reference buffers the row and its fit then fails -> NULL
		if (entry.length > max_features)
			throw InvalidInputException("anofox_stats fit_agg (HIP): at most %llu features are supported, got %llu", (unsigned long long)max_features,
			                            (unsigned long long)entry.length);
		if (state.shard < 0) state.shard = (int64_t)arenas.ShardOf(&state, entry.length); // routed once, at the first accepted row
		if (!writer || cur_p != entry.length || cur_shard != (uint32_t)state.shard) {
			cur_p = entry.length;
			cur_shard = (uint32_t)state.shard;
			cur_arena = &arenas.For(cur_p, cur_shard);
			auto &wslot = writers[std::make_pair(cur_p, cur_shard)];
			if (!wslot) wslot.reset(new anofox_shim::AggArena::Writer(*cur_arena));
			writer = wslot.get();
		}
		if (state.slot < 0) state.slot = writer->NewSlot();
		double *row = writer->Begin((uint32_t)state.slot, y_values[y_idx], entry.length, w);
		for (idx_t j = 0; j < entry.length; j++) // a NULL list element becomes NaN: the fit's row filter drops the row (ols.rs:59-66)
			row[j] = x_child_validity.RowIsValid(entry.offset + j) ? x_child_data[entry.offset + j] : NAN;
	}
}

// Combine: ols_aggregate.cpp:189-234.  A source without accepted rows is skipped.  A target without any adopts the
// source's slot when the source may be consumed (the reference moves the buffers), and otherwise gets a slot of its
// own that the source is merged into; pairs with slots on both sides are merged on the GPU, in the order given.
void HipAggCombine(Vector &source_vector, Vector &target_vector, AggregateInputData &aggr_input_data, idx_t count) {
	UnifiedVectorFormat source_data, target_data;
	source_vector.ToUnifiedFormat(count, source_data);
	target_vector.ToUnifiedFormat(count, target_data);
	auto sources = (HipAggState **)source_data.data;
	auto targets = (HipAggState **)target_data.data;
	auto &arenas = *aggr_input_data.bind_data->Cast<HipAggBindData>().arenas;
	const bool preserve = aggr_input_data.combine_type == AggregateCombineType::PRESERVE_INPUT;
	struct Pairs {
		vector<uint32_t> src, dst;
	};
	std::map<std::pair<size_t, uint32_t>, Pairs> local;                      // (width, shard): both sides on one device
	std::map<std::pair<size_t, std::pair<uint32_t, uint32_t>>, Pairs> cross; // (width, (source shard, target shard))
	for (idx_t i = 0; i < count; i++) {
		auto &source = *sources[source_data.sel->get_index(i)];
		auto &target = *targets[target_data.sel->get_index(i)];
		if (source.n_features < 0 || &source == &target) continue; // nothing to combine (:199-201)
		if (target.n_features < 0) {
			target.n_features = source.n_features;
			if (source.slot < 0) continue; // (rows with empty x lists only)
			if (!preserve) { // the reference moves the buffers: the target adopts the slot where it lives
				target.slot = source.slot;
				target.shard = source.shard;
				source.slot = -1;
				source.n_features = -1;
				source.shard = -1;
				continue;
			}
			target.shard = source.shard; // (an empty target has no home yet: it joins the source's device)
			target.slot = arenas.For((size_t)source.n_features, (uint32_t)target.shard).NewSlot();
		} else if (source.n_features != target.n_features) {
			throw InvalidInputException("Cannot combine states with different feature counts: %llu vs %llu", (unsigned long long)source.n_features,
			                            (unsigned long long)target.n_features); // :217-220
		}
		if (source.slot < 0) continue;
		if (target.slot < 0) { // a target that has only seen rows with empty x lists: it gets a slot next to the source
			target.shard = source.shard;
			target.slot = arenas.For((size_t)source.n_features, (uint32_t)target.shard).NewSlot();
		}
		const size_t p = (size_t)source.n_features;
		auto &pr = source.shard == target.shard ? local[std::make_pair(p, (uint32_t)source.shard)]
		                                        : cross[std::make_pair(p, std::make_pair((uint32_t)source.shard, (uint32_t)target.shard))];
		pr.src.push_back((uint32_t)source.slot);
		pr.dst.push_back((uint32_t)target.slot);
	}
	try {
		for (auto &kv : local) arenas.For(kv.first.first, kv.first.second).Combine(kv.second.src.data(), kv.second.dst.data(), kv.second.src.size(), preserve);
		// source and target on different devices: the sources' moment records travel (export -> import into fresh slots ->
		// the library's combine on the target's device); a consuming Combine then empties the sources where they were
		for (auto &kv : cross) {
			const size_t p = kv.first.first;
			auto &from = arenas.For(p, kv.first.second.first);
			auto &to = arenas.For(p, kv.first.second.second);
			vector<double> rec;
			vector<int64_t> cnt;
			from.ExportRecords(kv.second.src.data(), kv.second.src.size(), rec, cnt);
			to.EnsureState(p);
			to.MergeRecords(rec.data(), cnt.data(), kv.second.dst.data(), kv.second.dst.size());
			if (!preserve) from.ClearSlots(kv.second.src.data(), kv.second.src.size());
		}
	} catch (const std::runtime_error &e) {
		throw InvalidInputException(string(e.what()));
	}
}

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

// Finalize: ols_aggregate.cpp:249-338.  One arena call per feature count of the vector (one, normally): it fits what
// changed since it was last fitted (a GROUP BY: every group of the query, once) and returns this vector's records; they go
// into the STRUCT vector in the field order of the result type.  NULL where the reference returns NULL: fewer than 2
// accumulated rows (:263-267), a fit that failed (:298-301) — and a group the device state could not bring to the
// contract's accuracy (status 101).
void HipAggFinalize(Vector &state_vector, AggregateInputData &aggr_input_data, Vector &result, idx_t count, idx_t offset) {
	auto &bind = aggr_input_data.bind_data->Cast<HipAggBindData>();
	auto &arenas = *bind.arenas;
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipAggState **)sdata.data;
	struct Batch {
		vector<uint32_t> slots;
		vector<idx_t> rows;
		vector<int> status;
		vector<double> core, inf;
		string error;
	};
	std::map<std::pair<size_t, uint32_t>, Batch> batches; // (width, device shard)
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		if (state.slot < 0) {
			FlatVector::SetNull(result, i + offset, true); // no accepted row (or only rows with empty x lists)
			continue;
		}
		auto &b = batches[std::make_pair((size_t)state.n_features, (uint32_t)state.shard)];
		b.slots.push_back((uint32_t)state.slot);
		b.rows.push_back(i + offset);
	}
	const bool inference = bind.opts.compute_inference;
	auto &entries = StructVector::GetEntries(result);
	idx_t unrefined = 0;
	// one batched fit per device state; several devices fit concurrently (each arena has its own context, stream and lock)
	auto fetch = [&](const std::pair<size_t, uint32_t> &key, Batch &b) {
		const idx_t p = key.first;
		b.status.resize(b.slots.size());
		b.core.resize(b.slots.size() * (p + 6));
		b.inf.resize(inference ? b.slots.size() * (5 * p + 2) : 0);
		try {
			arenas.For(p, key.second).Fetch(b.slots.data(), b.slots.size(), b.core.data(), inference ? b.inf.data() : nullptr, b.status.data());
		} catch (const std::exception &e) {
			b.error = e.what();
		}
	};
	if (batches.size() <= 1) {
		for (auto &kv : batches) fetch(kv.first, kv.second);
	} else {
		vector<std::thread> workers;
		for (auto &kv : batches) workers.emplace_back([&fetch, &kv] { fetch(kv.first, kv.second); });
		for (auto &t : workers) t.join();
	}
	for (auto &kv : batches)
		if (!kv.second.error.empty()) throw InvalidInputException(kv.second.error);
	for (auto &kv : batches) {
		const idx_t p = kv.first.first;
		auto &b = kv.second;
		auto &status = b.status;
		auto &core = b.core;
		auto &inf = b.inf;
		for (idx_t k = 0; k < b.slots.size(); k++) {
			const idx_t r = b.rows[k];
			if (status[k] != 0) {
				if (status[k] == ANOFOX_HIP_STATUS_UNREFINED) unrefined++;
				FlatVector::SetNull(result, r, true);
				continue;
			}
			const double *rec = &core[k * (p + 6)];
			AppendList(*entries[0], r, rec, p);
			FlatVector::GetData<double>(*entries[1])[r] = rec[p];
			FlatVector::GetData<double>(*entries[2])[r] = rec[p + 1];
			FlatVector::GetData<double>(*entries[3])[r] = rec[p + 2];
			FlatVector::GetData<double>(*entries[4])[r] = rec[p + 3];
			FlatVector::GetData<int64_t>(*entries[5])[r] = (int64_t)rec[p + 4];
			FlatVector::GetData<int64_t>(*entries[6])[r] = (int64_t)p;
			if (inference) {
				const double *ir = &inf[k * (5 * p + 2)];
				for (idx_t f = 0; f < 5; f++) AppendList(*entries[7 + f], r, ir + f * p, p); // se, t, p, ci_lower, ci_upper
				FlatVector::GetData<double>(*entries[12])[r] = ir[5 * p];
				FlatVector::GetData<double>(*entries[13])[r] = ir[5 * p + 1];
			}
		}
	}
	// Groups the device state could neither resolve nor refit are NULL, never a number outside the contract; a site
	// that prefers a failing query sets ANOFOX_HIP_UNREFINED=error.  (HipAggStatsOf keeps the query's total.)
	if (unrefined) {
		const char *mode = getenv("ANOFOX_HIP_UNREFINED");
		if (mode && string(mode) == "error")
			throw InvalidInputException("anofox_stats fit_agg (HIP): %llu group(s) are ill-conditioned or fit (almost) exactly and their rows "
			                            "outgrew the row-log budgets (ANOFOX_HIP_RETAIN_BYTES / ANOFOX_HIP_RETAIN_HOST_BYTES)",
			                            (unsigned long long)unrefined);
	}
}

template <HipModel MODEL>
void RegisterHipAggregate(ExtensionLoader &loader, const char *name, const char *alias, const char *what, const char *example_opts) {
	constexpr bool kWeighted = MODEL == HipModel::WLS;
	vector<LogicalType> basic_args = {LogicalType::DOUBLE, LogicalType::LIST(LogicalType::DOUBLE)};
	vector<string> basic_names = {"y", "x"};
	if (kWeighted) {
		basic_args.push_back(LogicalType::DOUBLE);
		basic_names.push_back("weight");
	}
	vector<LogicalType> map_args = basic_args;
	map_args.push_back(LogicalType::ANY); // MAP or STRUCT of options, constant
	vector<string> map_names = basic_names;
	map_names.push_back("options");

	auto make = [&](const string &fname, const vector<LogicalType> &args) {
		return AggregateFunction(fname, args, LogicalType::ANY /* set in bind */, AggregateFunction::StateSize<HipAggState>, HipAggInitialize,
		                         HipAggUpdate<MODEL>, HipAggCombine, HipAggFinalize, nullptr /* simple_update */, HipAggBind<MODEL>, HipAggDestroy);
	};
	AggregateFunctionSet func_set(name);
	func_set.AddFunction(make(name, basic_args)); // (y, x[, weight]) — defaults
	func_set.AddFunction(make(name, map_args));   // (y, x[, weight], {'intercept': true, ...})
	CreateAggregateFunctionInfo info(std::move(func_set));
	info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
	FunctionDescription d1;
	d1.description = what;
	d1.examples = {string(name) + "(y, x" + (kWeighted ? ", w" : "") + ", " + example_opts + ")"};
	d1.categories = {"regression"};
	d1.parameter_names = map_names;
	d1.parameter_types = map_args;
	info.descriptions.push_back(std::move(d1));
	FunctionDescription d2;
	d2.description = what;
	d2.examples = {string(name) + "(y, x" + (kWeighted ? ", w" : "") + ")"};
	d2.categories = {"regression"};
	d2.parameter_names = basic_names;
	d2.parameter_types = basic_args;
	info.descriptions.push_back(std::move(d2));
	loader.RegisterFunction(std::move(info));

	AggregateFunctionSet alias_set(alias); // the short alias (ols_aggregate.cpp:415-425)
	alias_set.AddFunction(make(alias, basic_args));
	alias_set.AddFunction(make(alias, map_args));
	CreateAggregateFunctionInfo alias_info(std::move(alias_set));
	alias_info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
	alias_info.alias_of = name;
	loader.RegisterFunction(std::move(alias_info));
}

} // namespace

bool HipAggStatsOf(FunctionData &bind_data, HipAggStats &out) {
	auto *b = dynamic_cast<HipAggBindData *>(&bind_data);
	if (!b) return false;
	out = HipAggStats {};
	out.options = b->arenas->options;
	b->arenas->ForEach([&](size_t, anofox_shim::AggArena &a) {
		out.widths++;
		out.rows_accepted += a.RowsAccepted();
		out.unrefined += a.Unrefined();
		out.slot_high_water += a.SlotCount();
		out.live_slots += a.LiveSlots();
		out.fit_calls += a.FitCalls();
		out.slots_fitted += a.SlotsFitted();
	});
	return true;
}
const void *HipAggSharedStateOf(FunctionData &bind_data) {
	auto *b = dynamic_cast<HipAggBindData *>(&bind_data);
	return b ? (const void *)b->arenas.get() : nullptr;
}

void RegisterHipOlsAggregateFunction(ExtensionLoader &loader) {
	RegisterHipAggregate<HipModel::OLS>(loader, "anofox_stats_ols_fit_agg", "ols_fit_agg",
	                                    "Fits an OLS regression model and returns coefficients and fit statistics as a struct.", "{'fit_intercept': true}");
}
void RegisterHipRidgeAggregateFunction(ExtensionLoader &loader) {
	RegisterHipAggregate<HipModel::RIDGE>(loader, "anofox_stats_ridge_fit_agg", "ridge_fit_agg",
	                                      "Fits a Ridge regression model and returns coefficients and fit statistics as a struct.", "{'alpha': 1.0}");
}
void RegisterHipWlsAggregateFunction(ExtensionLoader &loader) {
	RegisterHipAggregate<HipModel::WLS>(loader, "anofox_stats_wls_fit_agg", "wls_fit_agg",
	                                    "Fits a weighted least squares model and returns coefficients and fit statistics as a struct.", "{'fit_intercept': true}");
}

} // namespace duckdb
