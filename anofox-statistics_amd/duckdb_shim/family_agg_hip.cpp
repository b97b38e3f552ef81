// family_agg_hip.cpp — DuckDB glue of the rest of the regression family over the batched C ABI:
//
//   {ols,ridge,wls}_fit_predict_agg   src/aggregate_functions/ols_predict_aggregate.cpp   (state :21-58, bind data :63-88,
//                                     result type :93-104, Update :134-268, Combine :271-327, Finalize :330-425, Bind
//                                     :430-492, registration :497-603), ridge_predict_aggregate.cpp, wls_predict_aggregate.cpp
//   {ols,ridge,wls}_fit_predict       src/window_functions/ols_fit_predict.cpp (state :21-52, Update :110-193, Combine
//                                     :196-243, Finalize :246-324, Bind :329-355, registration :360-409),
//                                     ridge_fit_predict.cpp, wls_fit_predict.cpp — aggregates DuckDB's window operator drives
//   vif_agg                           src/aggregate_functions/vif_aggregate.cpp (Update :50-95, Combine :97-129, Finalize
//                                     :144-185, registration :201-232)
//
// The reference fits ONE state per FFI call inside its Finalize loop.  Here the DuckDB state still buffers the group's
// rows on the host (these aggregates return every row, in arrival order, so the rows must be kept whatever fits them) and
// Finalize turns the whole vector of states (up to 2048) into ONE call of anofox_hip_fit_predict_batch_host /
// anofox_hip_vif_batch_host per feature count: states = groups, columns concatenated, NaN y = "does not train".
// The SQL surface is the reference's: names, aliases and deprecated aliases, the overloads with and without the split column
// and the constant options argument, LIST(STRUCT(y, yhat, yhat_lower, yhat_upper, is_training)) / STRUCT(yhat, yhat_lower,
// yhat_upper) / LIST(DOUBLE), NULL where the reference returns NULL.
//
// Compiled and driven in this repository against the stand-in of DuckDB's headers (tests/tools/duckdb_stub) with a mock of
// the C ABI under ASan / UBSan, and on the GPU with the real library against the oracle (tests/test_gpu_glue.py).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <memory>

#include "duckdb.hpp"
#include "duckdb/common/types/data_chunk.hpp"
#include "duckdb/execution/expression_executor.hpp"
#include "duckdb/function/aggregate_function.hpp"
#include "duckdb/main/extension/extension_loader.hpp"
#include "duckdb/parser/parsed_data/create_aggregate_function_info.hpp"

#include "anofox_stats_hip.h"
#include "family_agg_hip.hpp"
#include "hip_options.hpp"

namespace duckdb {

namespace {
using namespace hip_glue;

[[noreturn]] void ThrowAbi(const AnofoxError &err) {
	throw InvalidInputException("anofox_stats (HIP): %s", err.message[0] ? err.message : "the batched call failed");
}

// ---- bind data of the predict aggregate and the window aggregate: the parsed options ----
struct HipFamilyBindData : public FunctionData {
	HipFamilyBindData(HipModel model_p, const HipFitOptions &opts_p, bool use_split_col_p) : model(model_p), opts(opts_p), use_split_col(use_split_col_p) {}
	HipModel model;
	HipFitOptions opts;
	bool use_split_col;
	AnofoxHipBatchOptions Batch() const {
		HipFitOptions o = opts;
		o.compute_inference = false; // both aggregates fit with inference off (ols_predict_aggregate.cpp:356, ols_fit_predict.cpp:283)
		return MakeHipOptions(model, o);
	}
	unique_ptr<FunctionData> Copy() const override { return make_uniq<HipFamilyBindData>(model, opts, use_split_col); }
	bool Equals(const FunctionData &other_p) const override {
		auto &other = other_p.Cast<HipFamilyBindData>();
		return model == other.model && opts == other.opts && use_split_col == other.use_split_col;
	}
};

// the row buffer behind a DuckDB state: everything Update accepted, in arrival order
struct RowBuffer {
	idx_t n_features = 0;
	vector<double> y;        // NaN where y was NULL
	vector<double> x;        // row-major, n_features per row; a NULL list element is NaN
	vector<double> w;        // weighted models only
	vector<uint8_t> flags;   // per row: kYNull | kTraining
	idx_t n_training = 0;
	// the window aggregate: x of the last row Update saw (ols_fit_predict.cpp:29-31)
	vector<double> current_x;
	bool has_current_x = false;
	idx_t Rows() const { return y.size(); }
	void Append(const RowBuffer &o) {
		y.insert(y.end(), o.y.begin(), o.y.end());
		x.insert(x.end(), o.x.begin(), o.x.end());
		w.insert(w.end(), o.w.begin(), o.w.end());
		flags.insert(flags.end(), o.flags.begin(), o.flags.end());
		n_training += o.n_training;
	}
};
constexpr uint8_t kYNull = 1, kTraining = 2;

// The DuckDB-side state is one pointer: Initialize has nothing to construct and an untouched state costs no allocation
// (a window query makes a state per output row).
struct HipRowsState {
	RowBuffer *rows;
};

void HipRowsInitialize(const AggregateFunction &, data_ptr_t state_p) { reinterpret_cast<HipRowsState *>(state_p)->rows = nullptr; }

void HipRowsDestroy(Vector &state_vector, AggregateInputData &, idx_t count) {
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipRowsState **)sdata.data;
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		delete state.rows;
		state.rows = nullptr;
	}
}

RowBuffer &Rows(HipRowsState &state, idx_t n_features, bool message_with_counts) {
	if (!state.rows) {
		state.rows = new RowBuffer();
		state.rows->n_features = n_features; // the first row with a non-NULL x list fixes it (ols_predict_aggregate.cpp:182-187)
	}
	if (state.rows->n_features != n_features) {
		if (message_with_counts)
			throw InvalidInputException("Inconsistent feature count: expected %llu, got %llu", (unsigned long long)state.rows->n_features,
			                            (unsigned long long)n_features);
		throw InvalidInputException("Inconsistent feature count"); // the text of the ridge / wls files
	}
	return *state.rows;
}

bool IsSplitTraining(const string_t &split) { // 'train' / 'training', any case (ols_predict_aggregate.cpp:125-132)
	string v = split.GetString();
	for (auto &c : v) c = (char)std::tolower((unsigned char)c);
	return v == "train" || v == "training";
}

// Combine of both row-buffering aggregates: an initialised source is appended to the target; an uninitialised target takes the
// source's buffer (moved when the source may be consumed, copied under PRESERVE_INPUT — the window segment tree)
template <bool WINDOW>
void HipRowsCombine(Vector &source_vector, Vector &target_vector, AggregateInputData &aggr_input_data, idx_t count) {
	UnifiedVectorFormat source_data, target_data;
	source_vector.ToUnifiedFormat(count, source_data);
	target_vector.ToUnifiedFormat(count, target_data);
	auto sources = (HipRowsState **)source_data.data;
	auto targets = (HipRowsState **)target_data.data;
	const bool preserve = aggr_input_data.combine_type == AggregateCombineType::PRESERVE_INPUT;
	for (idx_t i = 0; i < count; i++) {
		auto &source = *sources[source_data.sel->get_index(i)];
		auto &target = *targets[target_data.sel->get_index(i)];
		if (!source.rows || &source == &target) continue;
		if (!target.rows) {
			if (preserve) {
				target.rows = new RowBuffer(*source.rows);
			} else {
				target.rows = source.rows;
				source.rows = nullptr;
			}
			continue;
		}
		if (source.rows->n_features != target.rows->n_features) {
			if (WINDOW) throw InvalidInputException("Cannot combine states with different feature counts");
			throw InvalidInputException("Cannot combine states with different feature counts: %llu vs %llu", (unsigned long long)source.rows->n_features,
			                            (unsigned long long)target.rows->n_features);
		}
		target.rows->Append(*source.rows);
		if (WINDOW && source.rows->has_current_x) { // the later state's row is the frame's last (ols_fit_predict.cpp:238-241)
			target.rows->current_x = source.rows->current_x;
			target.rows->has_current_x = true;
		}
	}
}

// ---- the states of one Finalize vector as one batch per feature count ----
struct FamilyBatch {
	idx_t p = 0;
	vector<idx_t> result_rows;     // where each group of the batch goes
	vector<RowBuffer *> buffers;
	vector<int64_t> offsets {0};
	vector<int64_t> train_counts;
	vector<double> y, w, cols;     // cols: p columns of n rows, one after the other
	vector<double> core, pred;
	// extra_row: the window aggregate appends its current x as a row that does not train
	void Run(const AnofoxHipBatchOptions &options, bool weighted, bool extra_row) {
		int64_t n = 0;
		for (auto *b : buffers) {
			n += (int64_t)b->Rows() + (extra_row ? 1 : 0);
			offsets.push_back(n);
			train_counts.push_back((int64_t)b->n_training);
		}
		y.resize((size_t)n);
		cols.resize((size_t)n * p);
		if (weighted) w.resize((size_t)n);
		int64_t at = 0;
		for (auto *b : buffers) {
			const idx_t rows = b->Rows();
			for (idx_t r = 0; r < rows; r++) {
				const bool train = b->flags[r] & kTraining;
				y[at + r] = train ? b->y[r] : NAN; // the batch ABI's "does not train"
				if (weighted) w[at + r] = train ? b->w[r] : 1.0;
				for (idx_t j = 0; j < p; j++) cols[j * (size_t)n + at + r] = b->x[r * p + j];
			}
			at += (int64_t)rows;
			if (extra_row) {
				y[at] = NAN;
				if (weighted) w[at] = 1.0;
				for (idx_t j = 0; j < p; j++) cols[j * (size_t)n + at] = b->current_x[j];
				at++;
			}
		}
		vector<const double *> col_ptrs(p);
		for (idx_t j = 0; j < p; j++) col_ptrs[j] = cols.data() + j * (size_t)n;
		core.resize(buffers.size() * (p + 6));
		pred.resize((size_t)n * 3);
		AnofoxError err;
		memset(&err, 0, sizeof err);
		if (!anofox_hip_fit_predict_batch_host(nullptr, (int64_t)buffers.size(), p, n, offsets.data(), y.data(), col_ptrs.data(),
		                                       weighted ? w.data() : nullptr, train_counts.data(), options, core.data(), pred.data(), &err))
			ThrowAbi(err);
	}
	bool Failed(idx_t g) const { return core[g * (p + 6) + p + 5] != 0.0; }
};

// =====================================================================================================================
// *_fit_predict_agg(y, x[, weights][, split_col][, options]) -> LIST(STRUCT(y, yhat, yhat_lower, yhat_upper, is_training))
// =====================================================================================================================
LogicalType GetHipPredictAggResultType() { // ols_predict_aggregate.cpp:93-104
	child_list_t<LogicalType> row_children;
	row_children.push_back(make_pair("y", LogicalType::DOUBLE));
	row_children.push_back(make_pair("yhat", LogicalType::DOUBLE));
	row_children.push_back(make_pair("yhat_lower", LogicalType::DOUBLE));
	row_children.push_back(make_pair("yhat_upper", LogicalType::DOUBLE));
	row_children.push_back(make_pair("is_training", LogicalType::BOOLEAN));
	return LogicalType::LIST(LogicalType::STRUCT(std::move(row_children)));
}

// Update: every row with a non-NULL x list (and, weighted, a non-NULL weight) is kept for the output; it trains iff y is not
// NULL (or the split column says so AND y is not NULL), no list element is NULL, the weight is positive, and — under
// null_policy = 'drop_y_zero_x' — no feature is exactly 0 (ols_predict_aggregate.cpp:134-268, wls_predict_aggregate.cpp:160-240)
template <HipModel MODEL>
void HipPredictAggUpdate(Vector inputs[], AggregateInputData &aggr_input_data, idx_t input_count, Vector &state_vector, idx_t count) {
	constexpr bool kWeighted = MODEL == HipModel::WLS;
	auto &bind = aggr_input_data.bind_data->Cast<HipFamilyBindData>();
	const idx_t split_idx_arg = kWeighted ? 3 : 2;
	if (input_count < (kWeighted ? 3u : 2u)) throw InvalidInputException("anofox_stats fit_predict_agg (HIP): too few arguments");
	UnifiedVectorFormat y_data, x_data, w_data, split_data, sdata;
	inputs[0].ToUnifiedFormat(count, y_data);
	inputs[1].ToUnifiedFormat(count, x_data);
	if (kWeighted) inputs[2].ToUnifiedFormat(count, w_data);
	auto y_values = UnifiedVectorFormat::GetData<double>(y_data);
	auto w_values = kWeighted ? UnifiedVectorFormat::GetData<double>(w_data) : nullptr;
	auto x_list = UnifiedVectorFormat::GetData<list_entry_t>(x_data);
	auto &x_child = ListVector::GetEntry(inputs[1]);
	auto x_child_data = FlatVector::GetData<double>(x_child);
	auto &x_child_validity = FlatVector::Validity(x_child);
	const string_t *split_values = nullptr;
	if (bind.use_split_col && input_count > split_idx_arg) {
		inputs[split_idx_arg].ToUnifiedFormat(count, split_data);
		split_values = UnifiedVectorFormat::GetData<string_t>(split_data);
	}
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipRowsState **)sdata.data;
	const idx_t max_features = anofox_hip_max_features();
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		auto x_idx = x_data.sel->get_index(i);
		if (!x_data.validity.RowIsValid(x_idx)) continue; // a NULL x list: the row does not exist for this aggregate
		double weight = 1.0;
		if (kWeighted) {
			auto w_idx = w_data.sel->get_index(i);
			if (!w_data.validity.RowIsValid(w_idx)) continue;
			weight = w_values[w_idx];
		}
		const auto entry = x_list[x_idx];
		if (entry.length > max_features)
			throw InvalidInputException("anofox_stats fit_predict_agg (HIP): at most %llu features are supported, got %llu",
			                            (unsigned long long)max_features, (unsigned long long)entry.length);
		auto &rows = Rows(state, entry.length, MODEL == HipModel::OLS);
		bool has_null_feature = false, has_zero = false;
		const size_t at = rows.x.size();
		rows.x.resize(at + entry.length);
		for (idx_t j = 0; j < entry.length; j++) {
			const idx_t pos = entry.offset + j;
			if (x_child_validity.RowIsValid(pos)) {
				rows.x[at + j] = x_child_data[pos];
				has_zero = has_zero || x_child_data[pos] == 0.0;
			} else {
				rows.x[at + j] = NAN; // never read the slot of a NULL (upstream issue #95)
				has_null_feature = true;
			}
		}
		auto y_idx = y_data.sel->get_index(i);
		const bool y_valid = y_data.validity.RowIsValid(y_idx);
		bool training = y_valid;
		if (bind.use_split_col && split_values) {
			auto s_idx = split_data.sel->get_index(i);
			training = split_data.validity.RowIsValid(s_idx) && IsSplitTraining(split_values[s_idx]) && y_valid;
		}
		if (kWeighted && !(weight > 0)) training = false; // wls_predict_aggregate.cpp:212-217 (`weight <= 0`; a NaN weight trains upstream and
		                                                    // fails the fit there — here it does not train)
		// a NULL feature: the OLS file clears the flag (ols_predict_aggregate.cpp:236-239); the ridge / wls files keep it — the row
		// is handed to the fit, whose row filter drops it — and so does the is_training column of their output
		if (has_null_feature && MODEL == HipModel::OLS) training = false;
		if (training && bind.opts.drop_y_zero_x && has_zero) training = false;
		rows.y.push_back(y_valid ? y_values[y_idx] : NAN);
		if (kWeighted) rows.w.push_back(weight);
		rows.flags.push_back((uint8_t)((y_valid ? 0 : kYNull) | (training ? kTraining : 0)));
		rows.n_training += training ? 1 : 0;
	}
}

void HipPredictAggFinalize(Vector &state_vector, AggregateInputData &aggr_input_data, Vector &result, idx_t count, idx_t offset) {
	auto &bind = aggr_input_data.bind_data->Cast<HipFamilyBindData>();
	const bool weighted = bind.model == HipModel::WLS;
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipRowsState **)sdata.data;
	std::map<idx_t, FamilyBatch> batches;
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		// fewer than 2 training rows -> NULL (ols_predict_aggregate.cpp:343-346); an empty x list cannot be fitted (the FFI refuses it)
		if (!state.rows || state.rows->n_training < 2 || state.rows->n_features == 0) {
			FlatVector::SetNull(result, i + offset, true);
			continue;
		}
		auto &b = batches[state.rows->n_features];
		b.p = state.rows->n_features;
		b.result_rows.push_back(i + offset);
		b.buffers.push_back(state.rows);
	}
	const auto options = bind.Batch();
	for (auto &kv : batches) kv.second.Run(options, weighted, false);

	auto list_data = ListVector::GetData(result);
	for (auto &kv : batches) {
		auto &b = kv.second;
		for (idx_t g = 0; g < b.buffers.size(); g++) {
			const idx_t r = b.result_rows[g];
			if (b.Failed(g)) { // the fit failed -> NULL (ols_predict_aggregate.cpp:366-369)
				FlatVector::SetNull(result, r, true);
				continue;
			}
			const RowBuffer &rows = *b.buffers[g];
			const idx_t n_rows = rows.Rows();
			const idx_t list_offset = ListVector::GetListSize(result);
			ListVector::Reserve(result, list_offset + n_rows);
			ListVector::SetListSize(result, list_offset + n_rows);
			list_data[r].offset = list_offset;
			list_data[r].length = n_rows;
			auto &fields = StructVector::GetEntries(ListVector::GetEntry(result)); // [y, yhat, yhat_lower, yhat_upper, is_training]
			const double *pred = &b.pred[(size_t)b.offsets[g] * 3];
			for (idx_t row = 0; row < n_rows; row++) {
				const idx_t at = list_offset + row;
				if (rows.flags[row] & kYNull) FlatVector::SetNull(*fields[0], at, true);
				else FlatVector::GetData<double>(*fields[0])[at] = rows.y[row];
				if (isfinite(pred[row * 3])) { // :404-413: a non-finite prediction is three NULLs
					for (idx_t k = 0; k < 3; k++) FlatVector::GetData<double>(*fields[1 + k])[at] = pred[row * 3 + k];
				} else {
					for (idx_t k = 0; k < 3; k++) FlatVector::SetNull(*fields[1 + k], at, true);
				}
				FlatVector::GetData<bool>(*fields[4])[at] = (rows.flags[row] & kTraining) != 0;
			}
		}
	}
	// (the reference's state.Reset() after a successful fit: Destroy frees the buffers here)
}

template <HipModel MODEL, bool SPLIT>
unique_ptr<FunctionData> HipPredictAggBind(ClientContext &context, AggregateFunction &function, vector<unique_ptr<Expression>> &arguments) {
	HipFitOptions opts;
	const idx_t opt_idx = (MODEL == HipModel::WLS ? 3 : 2) + (SPLIT ? 1 : 0); // ols_predict_aggregate.cpp:435 / :466
	if (arguments.size() > opt_idx && arguments[opt_idx]->IsFoldable()) ParseHipFitOptions(ExpressionExecutor::EvaluateScalar(context, *arguments[opt_idx]), opts);
	function.return_type = GetHipPredictAggResultType();
	return make_uniq<HipFamilyBindData>(MODEL, opts, SPLIT);
}

template <HipModel MODEL>
void RegisterHipPredictAggregate(ExtensionLoader &loader, const string &model_name, const char *what) {
	constexpr bool kWeighted = MODEL == HipModel::WLS;
	const string name = "anofox_stats_" + model_name + "_fit_predict_agg";
	vector<LogicalType> basic = {LogicalType::DOUBLE, LogicalType::LIST(LogicalType::DOUBLE)};
	vector<string> basic_names = {"y", "x"};
	if (kWeighted) {
		basic.push_back(LogicalType::DOUBLE);
		basic_names.push_back("weights");
	}
	auto with = [](vector<LogicalType> v, std::initializer_list<LogicalType> more) {
		for (auto &t : more) v.push_back(t);
		return v;
	};
	auto make = [&](const string &fname, const vector<LogicalType> &args, bool split) {
		return AggregateFunction(fname, args, LogicalType::ANY /* set in bind */, AggregateFunction::StateSize<HipRowsState>, HipRowsInitialize,
		                         HipPredictAggUpdate<MODEL>, HipRowsCombine<false>, HipPredictAggFinalize, nullptr,
		                         split ? HipPredictAggBind<MODEL, true> : HipPredictAggBind<MODEL, false>, HipRowsDestroy);
	};
	const vector<LogicalType> map_args = with(basic, {LogicalType::ANY}), split_args = with(basic, {LogicalType::VARCHAR}),
	                          split_map_args = with(basic, {LogicalType::VARCHAR, LogicalType::ANY});
	auto fill = [&](const string &fname) {
		AggregateFunctionSet set(fname);
		set.AddFunction(make(fname, basic, false));         // (y, x[, weights])
		set.AddFunction(make(fname, map_args, false));      // (y, x[, weights], {'null_policy': 'drop', ...})
		set.AddFunction(make(fname, split_args, true));     // (y, x[, weights], split_col)
		set.AddFunction(make(fname, split_map_args, true)); // (y, x[, weights], split_col, options)
		return set;
	};
	CreateAggregateFunctionInfo info(fill(name));
	info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
	const string head = name + "(y, x" + (kWeighted ? ", weights" : "");
	auto describe = [&](const string &example, vector<string> names, const vector<LogicalType> &types) {
		FunctionDescription d;
		d.description = what;
		d.examples = {example};
		d.categories = {"regression", "prediction"};
		d.parameter_names = std::move(names);
		d.parameter_types = types;
		info.descriptions.push_back(std::move(d));
	};
	auto names_with = [&](std::initializer_list<const char *> more) {
		vector<string> v = basic_names;
		for (auto m : more) v.push_back(m);
		return v;
	};
	describe(head + ")", names_with({}), basic);
	describe(head + ", {'null_policy': 'drop'})", names_with({"options"}), map_args);
	describe(head + ", split_col)", names_with({"split_col"}), split_args);
	describe(head + ", split_col, {'null_policy': 'drop'})", names_with({"split_col", "options"}), split_map_args);
	loader.RegisterFunction(std::move(info));
	// the short alias and the two deprecated names (ols_predict_aggregate.cpp:563-602)
	for (const string &alias : {model_name + "_fit_predict_agg", model_name + "_predict_agg", "anofox_stats_" + model_name + "_predict_agg"}) {
		CreateAggregateFunctionInfo alias_info(fill(alias));
		alias_info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
		alias_info.alias_of = name;
		loader.RegisterFunction(std::move(alias_info));
	}
}

// =====================================================================================================================
// *_fit_predict(y, x[, weight][, options]) OVER (...) -> STRUCT(yhat, yhat_lower, yhat_upper)
// =====================================================================================================================
LogicalType GetHipFitPredictResultType() { // ols_fit_predict.cpp:84-90
	child_list_t<LogicalType> children;
	children.push_back(make_pair("yhat", LogicalType::DOUBLE));
	children.push_back(make_pair("yhat_lower", LogicalType::DOUBLE));
	children.push_back(make_pair("yhat_upper", LogicalType::DOUBLE));
	return LogicalType::STRUCT(std::move(children));
}

// Update (ols_fit_predict.cpp:110-193): the frame's rows arrive one by one; the last one with a non-NULL x list is the row to
// predict, every one with a non-NULL y (and weight) trains.  Only training rows are buffered.
template <HipModel MODEL>
void HipFitPredictUpdate(Vector inputs[], AggregateInputData &aggr_input_data, idx_t input_count, Vector &state_vector, idx_t count) {
	constexpr bool kWeighted = MODEL == HipModel::WLS;
	auto &bind = aggr_input_data.bind_data->Cast<HipFamilyBindData>();
	if (input_count < (kWeighted ? 3u : 2u)) throw InvalidInputException("anofox_stats fit_predict (HIP): too few arguments");
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
	auto states = (HipRowsState **)sdata.data;
	const idx_t max_features = anofox_hip_max_features();
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		auto x_idx = x_data.sel->get_index(i);
		if (!x_data.validity.RowIsValid(x_idx)) {
			if (state.rows) state.rows->has_current_x = false; // :141-144
			continue;
		}
		const auto entry = x_list[x_idx];
		if (entry.length > max_features)
			throw InvalidInputException("anofox_stats fit_predict (HIP): at most %llu features are supported, got %llu", (unsigned long long)max_features,
			                            (unsigned long long)entry.length);
		auto &rows = Rows(state, entry.length, MODEL == HipModel::OLS);
		rows.current_x.resize(entry.length);
		bool has_zero = false;
		for (idx_t j = 0; j < entry.length; j++) { // (a NULL list element: NaN — the reference reads the slot as it is)
			const idx_t pos = entry.offset + j;
			rows.current_x[j] = x_child_validity.RowIsValid(pos) ? x_child_data[pos] : NAN;
			has_zero = has_zero || rows.current_x[j] == 0.0;
		}
		rows.has_current_x = true;
		auto y_idx = y_data.sel->get_index(i);
		bool training = y_data.validity.RowIsValid(y_idx);
		double weight = 1.0;
		if (kWeighted) {
			auto w_idx = w_data.sel->get_index(i);
			training = training && w_data.validity.RowIsValid(w_idx); // wls_fit_predict.cpp:150-154
			if (training) weight = w_values[w_idx];
		}
		if (training && bind.opts.drop_y_zero_x && has_zero) training = false;
		if (!training) continue;
		rows.y.push_back(y_values[y_idx]);
		rows.x.insert(rows.x.end(), rows.current_x.begin(), rows.current_x.end());
		if (kWeighted) rows.w.push_back(weight);
		rows.flags.push_back(kTraining);
		rows.n_training++;
	}
}

// Finalize (ols_fit_predict.cpp:246-324): NULL without a current row or with at most p + [intercept] training rows; otherwise
// the state's training rows plus its current row form one group of the batch, and the prediction of that last row is the result
void HipFitPredictFinalize(Vector &state_vector, AggregateInputData &aggr_input_data, Vector &result, idx_t count, idx_t offset) {
	auto &bind = aggr_input_data.bind_data->Cast<HipFamilyBindData>();
	const bool weighted = bind.model == HipModel::WLS;
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipRowsState **)sdata.data;
	std::map<idx_t, FamilyBatch> batches;
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		if (!state.rows || !state.rows->has_current_x || state.rows->n_features == 0) {
			FlatVector::SetNull(result, i + offset, true);
			continue;
		}
		const idx_t min_obs = state.rows->n_features + (bind.opts.fit_intercept ? 1 : 0);
		if (state.rows->n_training <= min_obs) {
			FlatVector::SetNull(result, i + offset, true);
			continue;
		}
		auto &b = batches[state.rows->n_features];
		b.p = state.rows->n_features;
		b.result_rows.push_back(i + offset);
		b.buffers.push_back(state.rows);
	}
	const auto options = bind.Batch();
	for (auto &kv : batches) kv.second.Run(options, weighted, true);
	auto &fields = StructVector::GetEntries(result);
	for (auto &kv : batches) {
		auto &b = kv.second;
		for (idx_t g = 0; g < b.buffers.size(); g++) {
			const idx_t r = b.result_rows[g];
			if (b.Failed(g)) {
				FlatVector::SetNull(result, r, true);
				continue;
			}
			// the current row is the group's last; a prediction that is not finite stays what the library made of it (NaN), as the
			// reference writes whatever anofox_predict_with_interval returned (:311-318)
			const double *pred = &b.pred[((size_t)b.offsets[g + 1] - 1) * 3];
			for (idx_t k = 0; k < 3; k++) FlatVector::GetData<double>(*fields[k])[r] = pred[k];
		}
	}
}

template <HipModel MODEL>
unique_ptr<FunctionData> HipFitPredictBind(ClientContext &context, AggregateFunction &function, vector<unique_ptr<Expression>> &arguments) {
	HipFitOptions opts;
	const idx_t opt_idx = MODEL == HipModel::WLS ? 3 : 2; // ols_fit_predict.cpp:333
	if (arguments.size() > opt_idx && arguments[opt_idx]->IsFoldable()) ParseHipFitOptions(ExpressionExecutor::EvaluateScalar(context, *arguments[opt_idx]), opts);
	function.return_type = GetHipFitPredictResultType();
	return make_uniq<HipFamilyBindData>(MODEL, opts, false);
}

template <HipModel MODEL>
void RegisterHipFitPredict(ExtensionLoader &loader, const string &model_name, const char *what) {
	constexpr bool kWeighted = MODEL == HipModel::WLS;
	const string name = "anofox_stats_" + model_name + "_fit_predict";
	vector<LogicalType> basic = {LogicalType::DOUBLE, LogicalType::LIST(LogicalType::DOUBLE)};
	vector<string> basic_names = {"y", "x"};
	if (kWeighted) {
		basic.push_back(LogicalType::DOUBLE);
		basic_names.push_back("weight");
	}
	vector<LogicalType> map_args = basic;
	map_args.push_back(LogicalType::ANY);
	vector<string> map_names = basic_names;
	map_names.push_back("options");
	auto fill = [&](const string &fname) {
		AggregateFunctionSet set(fname);
		for (auto *args : {&basic, &map_args})
			set.AddFunction(AggregateFunction(fname, *args, GetHipFitPredictResultType(), AggregateFunction::StateSize<HipRowsState>, HipRowsInitialize,
			                                  HipFitPredictUpdate<MODEL>, HipRowsCombine<true>, HipFitPredictFinalize, nullptr, HipFitPredictBind<MODEL>,
			                                  HipRowsDestroy));
		return set;
	};
	CreateAggregateFunctionInfo info(fill(name));
	info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
	const string head = name + "(y, x" + (kWeighted ? ", weight" : "");
	FunctionDescription d1;
	d1.description = what;
	d1.examples = {head + ")"};
	d1.categories = {"regression", "prediction"};
	d1.parameter_names = basic_names;
	d1.parameter_types = basic;
	info.descriptions.push_back(std::move(d1));
	FunctionDescription d2;
	d2.description = what;
	d2.examples = {head + ", {'null_policy': 'drop'})"};
	d2.categories = {"regression", "prediction"};
	d2.parameter_names = map_names;
	d2.parameter_types = map_args;
	info.descriptions.push_back(std::move(d2));
	loader.RegisterFunction(std::move(info));
	CreateAggregateFunctionInfo alias_info(fill(model_name + "_fit_predict"));
	alias_info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
	alias_info.alias_of = name;
	loader.RegisterFunction(std::move(alias_info));
}

// =====================================================================================================================
// vif_agg(x LIST(DOUBLE)) -> LIST(DOUBLE)
// =====================================================================================================================
// the state: one column of values per feature — Update appends every non-NaN value to ITS column (vif_aggregate.cpp:86-92),
// so a NaN shortens that column only and Finalize meets columns of unequal length (-> NULL, vif.rs:40-51)
struct VifColumns {
	vector<vector<double>> columns;
};
struct HipVifState {
	VifColumns *cols;
};

void HipVifInitialize(const AggregateFunction &, data_ptr_t state_p) { reinterpret_cast<HipVifState *>(state_p)->cols = nullptr; }

void HipVifDestroy(Vector &state_vector, AggregateInputData &, idx_t count) {
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipVifState **)sdata.data;
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		delete state.cols;
		state.cols = nullptr;
	}
}

void HipVifUpdate(Vector inputs[], AggregateInputData &, idx_t input_count, Vector &state_vector, idx_t count) {
	if (input_count < 1) throw InvalidInputException("anofox_stats vif_agg (HIP): too few arguments");
	UnifiedVectorFormat x_data, sdata;
	inputs[0].ToUnifiedFormat(count, x_data);
	auto x_list = UnifiedVectorFormat::GetData<list_entry_t>(x_data);
	auto &x_child = ListVector::GetEntry(inputs[0]);
	auto x_child_data = FlatVector::GetData<double>(x_child);
	auto &x_child_validity = FlatVector::Validity(x_child);
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipVifState **)sdata.data;
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		auto x_idx = x_data.sel->get_index(i);
		if (!x_data.validity.RowIsValid(x_idx)) continue;
		const auto entry = x_list[x_idx];
		if (!state.cols) {
			state.cols = new VifColumns();
			state.cols->columns.resize(entry.length);
		}
		auto &columns = state.cols->columns;
		if (entry.length != columns.size())
			throw InvalidInputException("Inconsistent feature count: expected %llu, got %llu", (unsigned long long)columns.size(),
			                            (unsigned long long)entry.length);
		for (idx_t j = 0; j < entry.length; j++) {
			const idx_t pos = entry.offset + j;
			if (!x_child_validity.RowIsValid(pos)) continue; // (a NULL element: as a NaN — the reference reads the slot as it is)
			const double v = x_child_data[pos];
			if (!std::isnan(v)) columns[j].push_back(v);
		}
	}
}

void HipVifCombine(Vector &source_vector, Vector &target_vector, AggregateInputData &aggr_input_data, idx_t count) {
	UnifiedVectorFormat source_data, target_data;
	source_vector.ToUnifiedFormat(count, source_data);
	target_vector.ToUnifiedFormat(count, target_data);
	auto sources = (HipVifState **)source_data.data;
	auto targets = (HipVifState **)target_data.data;
	const bool preserve = aggr_input_data.combine_type == AggregateCombineType::PRESERVE_INPUT;
	for (idx_t i = 0; i < count; i++) {
		auto &source = *sources[source_data.sel->get_index(i)];
		auto &target = *targets[target_data.sel->get_index(i)];
		if (!source.cols || &source == &target) continue;
		if (!target.cols) {
			if (preserve) {
				target.cols = new VifColumns(*source.cols);
			} else {
				target.cols = source.cols;
				source.cols = nullptr;
			}
			continue;
		}
		if (source.cols->columns.size() != target.cols->columns.size())
			throw InvalidInputException("Cannot combine states with different feature counts: %llu vs %llu", (unsigned long long)source.cols->columns.size(),
			                            (unsigned long long)target.cols->columns.size());
		for (idx_t j = 0; j < target.cols->columns.size(); j++)
			target.cols->columns[j].insert(target.cols->columns[j].end(), source.cols->columns[j].begin(), source.cols->columns[j].end());
	}
}

// Finalize (vif_aggregate.cpp:144-185): NULL without rows, with fewer than 2 features or fewer than 3 values in the first
// column, and when the columns differ in length; every other state of the vector goes into one batched call per feature count
void HipVifFinalize(Vector &state_vector, AggregateInputData &, Vector &result, idx_t count, idx_t offset) {
	UnifiedVectorFormat sdata;
	state_vector.ToUnifiedFormat(count, sdata);
	auto states = (HipVifState **)sdata.data;
	struct Batch {
		vector<idx_t> result_rows;
		vector<VifColumns *> cols;
	};
	std::map<idx_t, Batch> batches;
	const idx_t max_features = anofox_hip_vif_max_features();
	for (idx_t i = 0; i < count; i++) {
		auto &state = *states[sdata.sel->get_index(i)];
		bool usable = state.cols && state.cols->columns.size() >= 2 && state.cols->columns[0].size() >= 3;
		if (usable)
			for (auto &c : state.cols->columns) usable = usable && c.size() == state.cols->columns[0].size();
		if (!usable) {
			FlatVector::SetNull(result, i + offset, true);
			continue;
		}
		if (state.cols->columns.size() > max_features)
			throw InvalidInputException("anofox_stats vif_agg (HIP): at most %llu features are supported, got %llu", (unsigned long long)max_features,
			                            (unsigned long long)state.cols->columns.size());
		auto &b = batches[state.cols->columns.size()];
		b.result_rows.push_back(i + offset);
		b.cols.push_back(state.cols);
	}
	auto list_data = ListVector::GetData(result);
	for (auto &kv : batches) {
		const idx_t p = kv.first;
		auto &b = kv.second;
		vector<int64_t> offsets {0};
		for (auto *c : b.cols) offsets.push_back(offsets.back() + (int64_t)c->columns[0].size());
		const size_t n = (size_t)offsets.back();
		vector<double> cols(n * p);
		for (idx_t g = 0; g < b.cols.size(); g++)
			for (idx_t j = 0; j < p; j++) std::copy(b.cols[g]->columns[j].begin(), b.cols[g]->columns[j].end(), cols.begin() + j * n + (size_t)offsets[g]);
		vector<const double *> col_ptrs(p);
		for (idx_t j = 0; j < p; j++) col_ptrs[j] = cols.data() + j * n;
		const size_t rec = anofox_hip_vif_record_len(p);
		vector<double> vif(b.cols.size() * rec);
		AnofoxError err;
		memset(&err, 0, sizeof err);
		if (!anofox_hip_vif_batch_host(nullptr, (int64_t)b.cols.size(), p, (int64_t)n, offsets.data(), col_ptrs.data(), vif.data(), &err)) ThrowAbi(err);
		for (idx_t g = 0; g < b.cols.size(); g++) {
			const idx_t r = b.result_rows[g];
			if (vif[g * rec + p] != 0.0) {
				FlatVector::SetNull(result, r, true);
				continue;
			}
			const idx_t list_offset = ListVector::GetListSize(result);
			ListVector::Reserve(result, list_offset + p); // (the reference's SetListInResult omits this, vif_aggregate.cpp:132-141)
			auto child = FlatVector::GetData<double>(ListVector::GetEntry(result));
			for (idx_t j = 0; j < p; j++) child[list_offset + j] = vif[g * rec + j];
			list_data[r].offset = list_offset;
			list_data[r].length = p;
			ListVector::SetListSize(result, list_offset + p);
		}
	}
}

unique_ptr<FunctionData> HipVifBind(ClientContext &, AggregateFunction &function, vector<unique_ptr<Expression>> &) {
	function.return_type = LogicalType::LIST(LogicalType::DOUBLE);
	return nullptr; // no options, no bind data (vif_aggregate.cpp:190-195)
}

} // namespace

void RegisterHipOlsFitPredictAggregateFunction(ExtensionLoader &loader) {
	RegisterHipPredictAggregate<HipModel::OLS>(loader, "ols", "Fits OLS regression over a partition and returns per-row predictions with confidence intervals.");
}
void RegisterHipRidgeFitPredictAggregateFunction(ExtensionLoader &loader) {
	RegisterHipPredictAggregate<HipModel::RIDGE>(loader, "ridge", "Fits Ridge regression over a partition and returns per-row predictions with confidence intervals.");
}
void RegisterHipWlsFitPredictAggregateFunction(ExtensionLoader &loader) {
	RegisterHipPredictAggregate<HipModel::WLS>(loader, "wls", "Fits WLS regression over a partition using weights and returns per-row predictions.");
}
void RegisterHipOlsFitPredictFunction(ExtensionLoader &loader) {
	RegisterHipFitPredict<HipModel::OLS>(loader, "ols", "Fits an OLS model over a window partition and returns predictions for each row, including confidence intervals.");
}
void RegisterHipRidgeFitPredictFunction(ExtensionLoader &loader) {
	RegisterHipFitPredict<HipModel::RIDGE>(loader, "ridge", "Fits a Ridge regression model over a window partition and returns predictions for each row.");
}
void RegisterHipWlsFitPredictFunction(ExtensionLoader &loader) {
	RegisterHipFitPredict<HipModel::WLS>(loader, "wls", "Fits a WLS regression model over a window partition using per-row weights and returns predictions.");
}
void RegisterHipVifAggregateFunction(ExtensionLoader &loader) {
	auto make = [](const string &fname) {
		return AggregateFunction(fname, {LogicalType::LIST(LogicalType::DOUBLE)}, LogicalType::ANY /* set in bind */, AggregateFunction::StateSize<HipVifState>,
		                         HipVifInitialize, HipVifUpdate, HipVifCombine, HipVifFinalize, nullptr, HipVifBind, HipVifDestroy);
	};
	AggregateFunctionSet func_set("anofox_stats_vif_agg");
	func_set.AddFunction(make("anofox_stats_vif_agg"));
	CreateAggregateFunctionInfo info(std::move(func_set));
	info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
	FunctionDescription d1;
	d1.description = "Aggregate version of VIF: computes Variance Inflation Factor for each feature from a column of feature vectors.";
	d1.examples = {"anofox_stats_vif_agg(x)"};
	d1.categories = {"regression-diagnostics"};
	d1.parameter_names = {"x"};
	d1.parameter_types = {LogicalType::LIST(LogicalType::DOUBLE)};
	info.descriptions.push_back(std::move(d1));
	loader.RegisterFunction(std::move(info));
	AggregateFunctionSet alias_set("vif_agg");
	alias_set.AddFunction(make("vif_agg"));
	CreateAggregateFunctionInfo alias_info(std::move(alias_set));
	alias_info.on_conflict = OnCreateConflict::ALTER_ON_CONFLICT;
	alias_info.alias_of = "anofox_stats_vif_agg";
	loader.RegisterFunction(std::move(alias_info));
}

} // namespace duckdb
