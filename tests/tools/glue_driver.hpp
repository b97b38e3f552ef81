// glue_driver.hpp — drives duckdb_shim/fit_agg_hip.cpp the way DuckDB's operators drive an aggregate function, on the
// stand-in API of tests/tools/duckdb_stub:
//   GroupBy   a parallel hash aggregate: worker threads with thread-local states fed by Update vectors (optionally as
//             dictionary vectors, so the selection-vector paths are taken), Combine of the thread-local states into the
//             global ones (ALLOW_DESTRUCTIVE, pairs in vectors), Finalize vector by vector with a result offset, Destroy
//   Window    DuckDB's naive window aggregator (a frame per output row: states initialised, fed their frames' rows in
//             mixed Update batches, finalized and destroyed per vector of output rows) — the reference's
//             test/sql/comprehensive_tests.test:425-444 runs the aggregate like this
//   TreeWindow  the segment-tree aggregator's use of Combine: leaf states built once, combined (PRESERVE_INPUT) into a
//             fresh state per output row, the same leaf feeding many frames of one Combine call
// Test infrastructure: used by glue_sanitize.cpp (mock of the C ABI, ASan / UBSan) and glue_capi.cpp (the real library
// on a GPU, from tests/test_gpu_glue.py).
#pragma once
#include <ctype.h>
#include <math.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "duckdb.hpp"

#include "../../anofox-statistics_amd/duckdb_shim/agg_arena.hpp"
#include "../../anofox-statistics_amd/duckdb_shim/fit_agg_hip.hpp"

namespace glue_driver {
using namespace duckdb;

// "key=value;key=value" -> a STRUCT literal ({'k': v, ...}) or, as_map, a MAP {'k': v} with one value type
inline Value ParseOptionSpec(const std::string &spec, bool as_map) {
	child_list_t<Value> kids;
	size_t at = 0;
	while (at < spec.size()) {
		size_t end = spec.find(';', at);
		if (end == std::string::npos) end = spec.size();
		const std::string item = spec.substr(at, end - at);
		at = end + 1;
		const size_t eq = item.find('=');
		if (eq == std::string::npos) continue;
		const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
		char *e1 = nullptr, *e2 = nullptr;
		const long long iv = strtoll(v.c_str(), &e1, 10);
		const double dv = strtod(v.c_str(), &e2);
		if (v == "true" || v == "false") kids.push_back({k, Value::BOOLEAN(v == "true")});
		else if (v == "null") kids.push_back({k, Value(LogicalType(LogicalType::DOUBLE))});
		else if (!v.empty() && e1 && *e1 == 0) kids.push_back({k, Value::INTEGER((int32_t)iv)});
		else if (!v.empty() && e2 && *e2 == 0) kids.push_back({k, Value::DOUBLE(dv)});
		else kids.push_back({k, Value(v)});
	}
	if (!as_map) return Value::STRUCT(std::move(kids));
	bool all_num = true;
	for (auto &k : kids) all_num = all_num && k.second.type().id() != LogicalTypeId::VARCHAR;
	vector<Value> keys, vals;
	for (auto &k : kids) {
		keys.push_back(Value(k.first));
		vals.push_back(all_num ? (k.second.IsNull() ? Value(LogicalType(LogicalType::DOUBLE)) : Value::DOUBLE(k.second.GetValue<double>())) : Value(k.second.ToString()));
	}
	return Value::MAP(LogicalType::VARCHAR, all_num ? LogicalType::DOUBLE : LogicalType::VARCHAR, keys, vals);
}

struct Inputs { // n rows; x row-major with p values per row
	size_t n = 0, p = 0;
	const double *y = nullptr, *x = nullptr, *w = nullptr;
	const uint8_t *y_null = nullptr, *x_null = nullptr, *xe_null = nullptr, *w_null = nullptr; // row / row / n x p / row
	const uint32_t *x_len = nullptr;                                                            // optional LIST length per row (default p)
};

struct Records { // what Finalize wrote, per output row
	size_t p = 0;  // the widest row's feature count; a row of q < p features has its lists left-aligned in p-wide fields
	bool inference = false;
	std::vector<double> core, inf; // [rows x (p + 6)], [rows x (5 p + 2)]  (n_features at core[p + 5])
	std::vector<uint8_t> is_null;
};

class Query {
public:
	// fn_name: any registered name or alias; options_spec == nullptr: the overload without the options argument
	Query(const std::string &fn_name, const char *options_spec, bool as_map, bool foldable = true) {
		RegisterHipOlsAggregateFunction(loader_);
		RegisterHipRidgeAggregateFunction(loader_);
		RegisterHipWlsAggregateFunction(loader_);
		auto it = loader_.registered.find(fn_name);
		if (it == loader_.registered.end()) throw std::runtime_error("no such function: " + fn_name);
		weighted_ = fn_name.find("wls") != std::string::npos;
		const size_t n_args = (weighted_ ? 3 : 2) + (options_spec ? 1 : 0);
		const AggregateFunction *pick = nullptr;
		for (auto &f : it->second.functions.functions)
			if (f.arguments.size() == n_args) pick = &f;
		if (!pick) throw std::runtime_error("no overload with that many arguments");
		fn_.reset(new AggregateFunction(*pick));
		vector<unique_ptr<Expression>> args;
		args.push_back(make_uniq<Expression>(Value(), false)); // y: a column reference
		args.push_back(make_uniq<Expression>(Value(), false)); // x
		if (weighted_) args.push_back(make_uniq<Expression>(Value(), false));
		if (options_spec) args.push_back(make_uniq<Expression>(ParseOptionSpec(options_spec, as_map), foldable));
		bind_ = fn_->bind(context_, *fn_, args);
		if (fn_->return_type.id() != LogicalTypeId::STRUCT) throw std::runtime_error("bind did not set a STRUCT return type");
		inference_ = fn_->return_type.children().size() == 14;
	}
	const ExtensionLoader &Loader() const { return loader_; }
	const LogicalType &ReturnType() const { return fn_->return_type; }
	bool Inference() const { return inference_; }
	HipAggStats Stats() {
		HipAggStats st;
		if (!HipAggStatsOf(*bind_, st)) throw std::runtime_error("not a bind data object of the HIP aggregates");
		return st;
	}
	FunctionData &BindData() { return *bind_; }

	// ---- a parallel hash aggregate ----
	Records GroupBy(const Inputs &in, const uint32_t *key, size_t n_keys, int n_threads, size_t vector_size, bool dictionary) {
		if (n_threads < 1) n_threads = 1;
		std::vector<std::vector<data_ptr_t>> local(n_threads, std::vector<data_ptr_t>(n_keys, nullptr));
		std::vector<std::string> errors(n_threads);
		std::vector<std::unique_ptr<FunctionData>> binds;
		for (int t = 0; t < n_threads; ++t) binds.push_back(bind_->Copy()); // every thread works with a copy of the bind data
		auto worker = [&](int t) {
			try {
				ArenaAllocator alloc;
				AggregateInputData aid(binds[t].get(), alloc);
				size_t v = 0;
				for (size_t r0 = 0; r0 < in.n; r0 += vector_size, ++v) {
					if ((int)(v % (size_t)n_threads) != t) continue;
					const size_t cnt = std::min(vector_size, in.n - r0);
					std::vector<size_t> rows(cnt);
					for (size_t i = 0; i < cnt; ++i) rows[i] = r0 + i;
					std::vector<data_ptr_t> sp(cnt);
					for (size_t i = 0; i < cnt; ++i) {
						data_ptr_t &st = local[t][key[rows[i]]];
						if (!st) st = NewState();
						sp[i] = st;
					}
					UpdateRows(aid, in, rows, sp, dictionary);
				}
			} catch (const std::exception &e) {
				errors[t] = e.what();
			}
		};
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
		for (auto &t : th) t.join();
		ArenaAllocator alloc;
		AggregateInputData aid(bind_.get(), alloc, AggregateCombineType::ALLOW_DESTRUCTIVE);
		for (auto &e : errors)
			if (!e.empty()) { // (a failing query still destroys its states)
				for (auto &l : local) DestroyStates(aid, l, vector_size);
				throw std::runtime_error(e);
			}
		// Combine: thread-local states into the global ones, pairs in vectors
		std::vector<data_ptr_t> global(n_keys, nullptr);
		for (size_t k = 0; k < n_keys; ++k) global[k] = NewState(); // the global table has a state per key
		Records out;
		try {
			for (int t = 0; t < n_threads; ++t) {
				std::vector<data_ptr_t> s, d;
				for (size_t k = 0; k < n_keys; ++k)
					if (local[t][k]) {
						s.push_back(local[t][k]);
						d.push_back(global[k]);
					}
				for (size_t c0 = 0; c0 < s.size(); c0 += vector_size) {
					const size_t cnt = std::min(vector_size, s.size() - c0);
					Vector sv = PointerVector(s.data() + c0, cnt), dv = PointerVector(d.data() + c0, cnt);
					fn_->combine(sv, dv, aid, cnt);
				}
			}
			out = FinalizeStates(aid, global, vector_size);
		} catch (...) {
			for (auto &l : local) DestroyStates(aid, l, vector_size);
			DestroyStates(aid, global, vector_size);
			throw;
		}
		for (auto &l : local) DestroyStates(aid, l, vector_size);
		DestroyStates(aid, global, vector_size);
		return out;
	}

	// ---- the naive window aggregator: ROWS BETWEEN `preceding` PRECEDING AND CURRENT ROW over the rows in order ----
	Records Window(const Inputs &in, size_t preceding, size_t vector_size) {
		ArenaAllocator alloc;
		AggregateInputData aid(bind_.get(), alloc);
		Records all;
		all.inference = inference_;
		for (size_t o0 = 0; o0 < in.n; o0 += vector_size) {
			const size_t cnt = std::min(vector_size, in.n - o0);
			std::vector<data_ptr_t> st(cnt);
			for (auto &s : st) s = NewState();
			std::vector<size_t> rows;
			std::vector<data_ptr_t> sp;
			for (size_t i = 0; i < cnt; ++i) {
				const size_t o = o0 + i;
				for (size_t r = o >= preceding ? o - preceding : 0; r <= o; ++r) {
					rows.push_back(r);
					sp.push_back(st[i]);
					if (rows.size() == vector_size) {
						UpdateRows(aid, in, rows, sp, true);
						rows.clear();
						sp.clear();
					}
				}
			}
			if (!rows.empty()) UpdateRows(aid, in, rows, sp, true);
			Records part = FinalizeStates(aid, st, vector_size);
			Append(all, part);
			DestroyStates(aid, st, vector_size);
		}
		return all;
	}

	// ---- the segment tree's Combine: leaves of `leaf` rows, frames of whole leaves [o / leaf - back, o / leaf] ----
	Records TreeWindow(const Inputs &in, size_t leaf, size_t back, size_t vector_size) {
		ArenaAllocator alloc;
		AggregateInputData aid(bind_.get(), alloc, AggregateCombineType::PRESERVE_INPUT);
		const size_t n_leaves = (in.n + leaf - 1) / leaf;
		std::vector<data_ptr_t> leaves(n_leaves);
		for (auto &s : leaves) s = NewState();
		for (size_t l = 0; l < n_leaves; ++l) {
			std::vector<size_t> rows;
			std::vector<data_ptr_t> sp;
			for (size_t r = l * leaf; r < std::min(in.n, (l + 1) * leaf); ++r) {
				rows.push_back(r);
				sp.push_back(leaves[l]);
			}
			UpdateRows(aid, in, rows, sp, false);
		}
		Records all;
		all.inference = inference_;
		for (size_t o0 = 0; o0 < n_leaves; o0 += vector_size) {
			const size_t cnt = std::min(vector_size, n_leaves - o0);
			std::vector<data_ptr_t> st(cnt);
			for (auto &s : st) s = NewState();
			std::vector<data_ptr_t> s, d;
			for (size_t i = 0; i < cnt; ++i) {
				const size_t o = o0 + i;
				for (size_t l = o >= back ? o - back : 0; l <= o; ++l) {
					s.push_back(leaves[l]);
					d.push_back(st[i]);
				}
			}
			for (size_t c0 = 0; c0 < s.size(); c0 += vector_size) {
				const size_t c = std::min(vector_size, s.size() - c0);
				Vector sv = PointerVector(s.data() + c0, c), dv = PointerVector(d.data() + c0, c);
				fn_->combine(sv, dv, aid, c);
			}
			Records part = FinalizeStates(aid, st, vector_size);
			Append(all, part);
			DestroyStates(aid, st, vector_size);
		}
		DestroyStates(aid, leaves, vector_size);
		return all;
	}

private:
	data_ptr_t NewState() {
		data_ptr_t s = new data_t[fn_->state_size(*fn_)];
		fn_->initialize(*fn_, s);
		return s;
	}
	static void FreeStates(std::vector<data_ptr_t> &v) {
		for (auto &s : v) {
			delete[] s;
			s = nullptr;
		}
	}
	static Vector PointerVector(data_ptr_t *ptrs, size_t cnt) {
		Vector v(LogicalType(LogicalType::POINTER), cnt);
		memcpy(FlatVector::GetData<data_ptr_t>(v), ptrs, cnt * sizeof(data_ptr_t));
		return v;
	}
	void DestroyStates(AggregateInputData &aid, std::vector<data_ptr_t> &states, size_t vector_size) {
		std::vector<data_ptr_t> live;
		for (auto s : states)
			if (s) live.push_back(s);
		for (size_t c0 = 0; c0 < live.size(); c0 += vector_size) {
			const size_t cnt = std::min(vector_size, live.size() - c0);
			Vector sv = PointerVector(live.data() + c0, cnt);
			fn_->destructor(sv, aid, cnt);
		}
		FreeStates(states);
	}
	// one Update call over the given input rows; dictionary: the inputs arrive as dictionary vectors over shuffled data
	void UpdateRows(AggregateInputData &aid, const Inputs &in, const std::vector<size_t> &rows, std::vector<data_ptr_t> &states, bool dictionary) {
		const size_t cnt = rows.size();
		if (cnt == 0) return;
		// physical order of the vector data: reversed when `dictionary`, with a selection that restores the row order
		std::vector<uint32_t> sel(cnt);
		for (size_t i = 0; i < cnt; ++i) sel[i] = (uint32_t)(dictionary ? cnt - 1 - i : i);
		std::vector<Vector> inputs;
		inputs.emplace_back(LogicalType(LogicalType::DOUBLE), cnt);
		inputs.emplace_back(LogicalType::LIST(LogicalType::DOUBLE), cnt);
		if (weighted_) inputs.emplace_back(LogicalType(LogicalType::DOUBLE), cnt);
		if (fn_->arguments.size() > (weighted_ ? 3u : 2u)) inputs.emplace_back(LogicalType(LogicalType::BIGINT), cnt); // the options constant
		double *yv = FlatVector::GetData<double>(inputs[0]);
		list_entry_t *le = ListVector::GetData(inputs[1]);
		Vector &child = ListVector::GetEntry(inputs[1]);
		size_t total = 0;
		for (size_t i = 0; i < cnt; ++i) total += in.x_len ? in.x_len[rows[i]] : in.p;
		ListVector::Reserve(inputs[1], total ? total : 1);
		double *cv = FlatVector::GetData<double>(child);
		size_t off = 0;
		for (size_t i = 0; i < cnt; ++i) {
			const size_t r = rows[i], phys = sel[i];
			yv[phys] = in.y[r];
			if (in.y_null && in.y_null[r]) FlatVector::SetNull(inputs[0], phys, true);
			const size_t len = in.x_len ? in.x_len[r] : in.p;
			le[phys].offset = off;
			le[phys].length = len;
			for (size_t j = 0; j < len; ++j) {
				cv[off + j] = j < in.p ? in.x[r * in.p + j] : 0.0;
				if (in.xe_null && j < in.p && in.xe_null[r * in.p + j]) FlatVector::Validity(child).SetInvalid(off + j);
			}
			off += len;
			if (in.x_null && in.x_null[r]) FlatVector::SetNull(inputs[1], phys, true);
			if (weighted_) {
				FlatVector::GetData<double>(inputs[2])[phys] = in.w ? in.w[r] : 1.0;
				if (in.w_null && in.w_null[r]) FlatVector::SetNull(inputs[2], phys, true);
			}
		}
		ListVector::SetListSize(inputs[1], total);
		if (dictionary)
			for (size_t k = 0; k < (weighted_ ? 3u : 2u); ++k) inputs[k].MakeDictionary(sel);
		if (inputs.size() > (weighted_ ? 3u : 2u)) inputs.back().MakeConstant();
		Vector sv = PointerVector(states.data(), cnt);
		fn_->update(inputs.data(), aid, inputs.size(), sv, cnt);
	}
	Records FinalizeStates(AggregateInputData &aid, std::vector<data_ptr_t> &states, size_t vector_size) {
		const size_t n = states.size();
		Vector result(fn_->return_type, n ? n : 1);
		for (size_t c0 = 0; c0 < n; c0 += vector_size) {
			const size_t cnt = std::min(vector_size, n - c0);
			Vector sv = PointerVector(states.data() + c0, cnt);
			fn_->finalize(sv, aid, result, cnt, c0);
		}
		Records out;
		out.inference = inference_;
		out.is_null.assign(n, 0);
		auto &entries = StructVector::GetEntries(result);
		size_t p = 0;
		for (size_t r = 0; r < n; ++r)
			if (FlatVector::Validity(result).RowIsValid(r)) p = std::max<size_t>(p, ListVector::GetData(*entries[0])[r].length);
		out.p = p;
		out.core.assign(n * (p + 6), NAN);
		if (inference_) out.inf.assign(n * (5 * p + 2), NAN);
		for (size_t r = 0; r < n; ++r) {
			if (!FlatVector::Validity(result).RowIsValid(r)) {
				out.is_null[r] = 1;
				continue;
			}
			double *c = &out.core[r * (p + 6)];
			const size_t q = (size_t)FlatVector::GetData<int64_t>(*entries[6])[r]; // this row's n_features
			auto copy_list = [&](Vector &lv, double *dst) {
				const list_entry_t e = ListVector::GetData(lv)[r];
				if (e.length != q || q > p || e.offset + e.length > ListVector::GetListSize(lv)) throw std::runtime_error("finalize wrote a bad LIST entry");
				memcpy(dst, FlatVector::GetData<double>(ListVector::GetEntry(lv)) + e.offset, q * sizeof(double));
			};
			copy_list(*entries[0], c);
			for (int k = 0; k < 4; ++k) c[p + k] = FlatVector::GetData<double>(*entries[1 + k])[r];
			c[p + 4] = (double)FlatVector::GetData<int64_t>(*entries[5])[r];
			c[p + 5] = (double)FlatVector::GetData<int64_t>(*entries[6])[r];
			if (inference_) {
				double *f = &out.inf[r * (5 * p + 2)];
				for (int k = 0; k < 5; ++k) copy_list(*entries[7 + k], f + k * p);
				f[5 * p] = FlatVector::GetData<double>(*entries[12])[r];
				f[5 * p + 1] = FlatVector::GetData<double>(*entries[13])[r];
			}
		}
		return out;
	}
	static void Append(Records &all, const Records &part) {
		if (all.core.empty() && all.is_null.empty()) all.p = part.p;
		if (part.p != all.p && part.p != 0) {
			if (all.p != 0) throw std::runtime_error("window parts with different widths");
			// earlier parts were all NULL: re-layout them for this width
			all.core.assign(all.is_null.size() * (part.p + 6), NAN);
			if (all.inference) all.inf.assign(all.is_null.size() * (5 * part.p + 2), NAN);
			all.p = part.p;
		}
		const size_t p = all.p, rows = part.is_null.size();
		for (size_t r = 0; r < rows; ++r) {
			all.is_null.push_back(part.is_null[r]);
			for (size_t j = 0; j < p + 6; ++j) all.core.push_back(part.p == p && !part.is_null[r] ? part.core[r * (p + 6) + j] : NAN);
			if (all.inference)
				for (size_t j = 0; j < 5 * p + 2; ++j) all.inf.push_back(part.p == p && !part.is_null[r] ? part.inf[r * (5 * p + 2) + j] : NAN);
		}
	}

	ExtensionLoader loader_;
	ClientContext context_;
	std::unique_ptr<AggregateFunction> fn_;
	unique_ptr<FunctionData> bind_;
	bool weighted_ = false, inference_ = false;
};

} // namespace glue_driver
