// family_driver.hpp — drives duckdb_shim/family_agg_hip.cpp the way DuckDB's operators drive an aggregate function, on the
// stand-in API of tests/tools/duckdb_stub (see glue_driver.hpp for the fit aggregates):
//   GroupBy     a parallel hash aggregate — worker threads with thread-local states fed by Update vectors (optionally as
//               dictionary vectors), Combine (ALLOW_DESTRUCTIVE) into the global states, Finalize vector by vector, Destroy —
//               for *_fit_predict_agg (with and without the split column) and vif_agg
//   Window      the naive window aggregator over ROWS BETWEEN k PRECEDING AND CURRENT ROW for *_fit_predict
//   TreeWindow  the segment tree's use of Combine (PRESERVE_INPUT): leaf states combined into a fresh state per output row
// A group's output rows come out in the order its states were combined: thread by thread, and within a thread in input
// order (the driver hands vector v to thread v % n_threads) — the tests reproduce that order.
// Test infrastructure: used by glue_sanitize.cpp (mock ABI, ASan / UBSan / TSan) and glue_capi.cpp (the real library on a GPU).
#pragma once
#include "glue_driver.hpp"

#include "../../anofox-statistics_amd/duckdb_shim/family_agg_hip.hpp"

namespace glue_driver {

// the strings behind the split codes of the tests: 0 = SQL NULL
static const char *const kSplitStrings[] = {nullptr, "train", "Training", "test", "TRAIN", "a-validation-partition-name", "training"};
constexpr size_t kSplitStringCount = sizeof(kSplitStrings) / sizeof(kSplitStrings[0]);

struct FamilyOut {
	std::vector<uint8_t> is_null;   // per state (group / output row)
	std::vector<int64_t> offsets;   // per state: its slice of the row arrays (predict_agg) or of vals (vif_agg); window: unused
	std::vector<double> vals;       // predict_agg: 4 per row {y, yhat, yhat_lower, yhat_upper}; window: 3 per state; vif: the lists
	std::vector<uint8_t> flags;     // predict_agg, per row: 1 y NULL, 2 / 4 / 8 yhat / lower / upper NULL, 16 is_training
};

class FamilyQuery {
public:
	enum class Kind { PREDICT_AGG, WINDOW, VIF };
	FamilyQuery(const std::string &fn_name, const char *options_spec, bool as_map, bool with_split) : with_split_(with_split) {
		RegisterHipOlsFitPredictAggregateFunction(loader_);
		RegisterHipRidgeFitPredictAggregateFunction(loader_);
		RegisterHipWlsFitPredictAggregateFunction(loader_);
		RegisterHipOlsFitPredictFunction(loader_);
		RegisterHipRidgeFitPredictFunction(loader_);
		RegisterHipWlsFitPredictFunction(loader_);
		RegisterHipVifAggregateFunction(loader_);
		auto it = loader_.registered.find(fn_name);
		if (it == loader_.registered.end()) throw std::runtime_error("no such function: " + fn_name);
		kind_ = fn_name.find("vif") != std::string::npos ? Kind::VIF : (fn_name.find("_agg") != std::string::npos ? Kind::PREDICT_AGG : Kind::WINDOW);
		weighted_ = fn_name.find("wls") != std::string::npos;
		if (with_split && kind_ != Kind::PREDICT_AGG) throw std::runtime_error("only the predict aggregates take a split column");
		vector<LogicalType> want;
		if (kind_ != Kind::VIF) want.push_back(LogicalType::DOUBLE);
		want.push_back(LogicalType::LIST(LogicalType::DOUBLE));
		if (weighted_) want.push_back(LogicalType::DOUBLE);
		if (with_split) want.push_back(LogicalType::VARCHAR);
		if (options_spec) want.push_back(LogicalType::ANY);
		const AggregateFunction *pick = nullptr;
		for (auto &f : it->second.functions.functions)
			if (f.arguments == want) pick = &f;
		if (!pick) throw std::runtime_error("no overload with these argument types");
		fn_.reset(new AggregateFunction(*pick));
		vector<unique_ptr<Expression>> args;
		for (size_t k = 0; k + (options_spec ? 1 : 0) < want.size(); ++k) args.push_back(make_uniq<Expression>(Value(), false)); // column references
		if (options_spec) args.push_back(make_uniq<Expression>(ParseOptionSpec(options_spec, as_map), true));
		bind_ = fn_->bind(context_, *fn_, args);
		const auto id = fn_->return_type.id();
		if (kind_ == Kind::WINDOW ? id != LogicalTypeId::STRUCT : id != LogicalTypeId::LIST) throw std::runtime_error("bind did not set the return type");
	}
	const ExtensionLoader &Loader() const { return loader_; }
	const LogicalType &ReturnType() const { return fn_->return_type; }
	Kind kind() const { return kind_; }

	// split: one code per row into kSplitStrings, or nullptr
	FamilyOut GroupBy(const Inputs &in, const uint8_t *split, const uint32_t *key, size_t n_keys, int n_threads, size_t vector_size, bool dictionary) {
		if (n_threads < 1) n_threads = 1;
		std::vector<std::vector<data_ptr_t>> local(n_threads, std::vector<data_ptr_t>(n_keys, nullptr));
		std::vector<std::string> errors(n_threads);
		std::vector<std::unique_ptr<FunctionData>> binds;
		for (int t = 0; t < n_threads; ++t) {
			binds.emplace_back();
			if (bind_) binds.back() = bind_->Copy(); // (vif_agg binds nothing)
		}
		auto worker = [&](int t) {
			try {
				ArenaAllocator alloc;
				AggregateInputData aid(binds[t].get(), alloc);
				size_t v = 0;
				for (size_t r0 = 0; r0 < in.n; r0 += vector_size, ++v) {
					if ((int)(v % (size_t)n_threads) != t) continue;
					const size_t cnt = std::min(vector_size, in.n - r0);
					std::vector<size_t> rows(cnt);
					std::vector<data_ptr_t> sp(cnt);
					for (size_t i = 0; i < cnt; ++i) {
						rows[i] = r0 + i;
						data_ptr_t &st = local[t][key[r0 + i]];
						if (!st) st = NewState();
						sp[i] = st;
					}
					UpdateRows(aid, in, split, rows, sp, dictionary);
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
			if (!e.empty()) {
				for (auto &l : local) DestroyStates(aid, l, vector_size);
				throw std::runtime_error(e);
			}
		std::vector<data_ptr_t> global(n_keys, nullptr);
		for (size_t k = 0; k < n_keys; ++k) global[k] = NewState();
		FamilyOut out;
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

	// ROWS BETWEEN `preceding` PRECEDING AND CURRENT ROW (preceding = SIZE_MAX: UNBOUNDED PRECEDING), a state per output row
	FamilyOut Window(const Inputs &in, size_t preceding, size_t vector_size) {
		ArenaAllocator alloc;
		AggregateInputData aid(bind_.get(), alloc);
		FamilyOut all;
		for (size_t o0 = 0; o0 < in.n; o0 += vector_size) {
			const size_t cnt = std::min(vector_size, in.n - o0);
			std::vector<data_ptr_t> st(cnt);
			for (auto &s : st) s = NewState();
			std::vector<size_t> rows;
			std::vector<data_ptr_t> sp;
			try {
				for (size_t i = 0; i < cnt; ++i) {
					const size_t o = o0 + i;
					for (size_t r = o >= preceding ? o - preceding : 0; r <= o; ++r) {
						rows.push_back(r);
						sp.push_back(st[i]);
						if (rows.size() == vector_size) {
							UpdateRows(aid, in, nullptr, rows, sp, true);
							rows.clear();
							sp.clear();
						}
					}
				}
				if (!rows.empty()) UpdateRows(aid, in, nullptr, rows, sp, true);
				Append(all, FinalizeStates(aid, st, vector_size));
			} catch (...) {
				DestroyStates(aid, st, vector_size);
				throw;
			}
			DestroyStates(aid, st, vector_size);
		}
		return all;
	}

	// leaves of `leaf` rows; output row o (one per leaf) = the leaves [o - back, o] combined, in order, into a fresh state
	FamilyOut TreeWindow(const Inputs &in, size_t leaf, size_t back, size_t vector_size) {
		ArenaAllocator alloc;
		AggregateInputData aid(bind_.get(), alloc, AggregateCombineType::PRESERVE_INPUT);
		const size_t n_leaves = (in.n + leaf - 1) / leaf;
		std::vector<data_ptr_t> leaves(n_leaves);
		for (auto &s : leaves) s = NewState();
		FamilyOut all;
		try {
			for (size_t l = 0; l < n_leaves; ++l) {
				std::vector<size_t> rows;
				std::vector<data_ptr_t> sp;
				for (size_t r = l * leaf; r < std::min(in.n, (l + 1) * leaf); ++r) {
					rows.push_back(r);
					sp.push_back(leaves[l]);
				}
				UpdateRows(aid, in, nullptr, rows, sp, false);
			}
			for (size_t o0 = 0; o0 < n_leaves; o0 += vector_size) {
				const size_t cnt = std::min(vector_size, n_leaves - o0);
				std::vector<data_ptr_t> st(cnt);
				for (auto &s : st) s = NewState();
				try {
					// frame by frame in leaf order: a Combine call holds at most one pair per target, as the segment tree's do
					for (size_t step = 0; step <= back; ++step) {
						std::vector<data_ptr_t> s, d;
						for (size_t i = 0; i < cnt; ++i) {
							const size_t o = o0 + i, first = o >= back ? o - back : 0;
							if (first + step > o) continue;
							s.push_back(leaves[first + step]);
							d.push_back(st[i]);
						}
						for (size_t c0 = 0; c0 < s.size(); c0 += vector_size) {
							const size_t c = std::min(vector_size, s.size() - c0);
							Vector sv = PointerVector(s.data() + c0, c), dv = PointerVector(d.data() + c0, c);
							fn_->combine(sv, dv, aid, c);
						}
					}
					Append(all, FinalizeStates(aid, st, vector_size));
				} catch (...) {
					DestroyStates(aid, st, vector_size);
					throw;
				}
				DestroyStates(aid, st, vector_size);
			}
		} catch (...) {
			DestroyStates(aid, leaves, vector_size);
			throw;
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
		for (auto &s : states) {
			delete[] s;
			s = nullptr;
		}
	}
	void UpdateRows(AggregateInputData &aid, const Inputs &in, const uint8_t *split, const std::vector<size_t> &rows, std::vector<data_ptr_t> &states,
	                bool dictionary) {
		const size_t cnt = rows.size();
		if (cnt == 0) return;
		std::vector<uint32_t> sel(cnt);
		for (size_t i = 0; i < cnt; ++i) sel[i] = (uint32_t)(dictionary ? cnt - 1 - i : i);
		std::vector<Vector> inputs;
		const size_t y_at = kind_ == Kind::VIF ? SIZE_MAX : 0, x_at = kind_ == Kind::VIF ? 0 : 1;
		if (kind_ != Kind::VIF) inputs.emplace_back(LogicalType(LogicalType::DOUBLE), cnt);
		inputs.emplace_back(LogicalType::LIST(LogicalType::DOUBLE), cnt);
		const size_t w_at = weighted_ ? inputs.size() : SIZE_MAX;
		if (weighted_) inputs.emplace_back(LogicalType(LogicalType::DOUBLE), cnt);
		const size_t s_at = with_split_ ? inputs.size() : SIZE_MAX;
		if (with_split_) inputs.emplace_back(LogicalType(LogicalType::VARCHAR), cnt);
		const size_t data_inputs = inputs.size();
		if (fn_->arguments.size() > data_inputs) inputs.emplace_back(LogicalType(LogicalType::BIGINT), cnt); // the options constant
		list_entry_t *le = ListVector::GetData(inputs[x_at]);
		Vector &child = ListVector::GetEntry(inputs[x_at]);
		size_t total = 0;
		for (size_t i = 0; i < cnt; ++i) total += in.x_len ? in.x_len[rows[i]] : in.p;
		ListVector::Reserve(inputs[x_at], total ? total : 1);
		double *cv = FlatVector::GetData<double>(child);
		size_t off = 0;
		for (size_t i = 0; i < cnt; ++i) {
			const size_t r = rows[i], phys = sel[i];
			if (y_at != SIZE_MAX) {
				FlatVector::GetData<double>(inputs[y_at])[phys] = in.y[r];
				if (in.y_null && in.y_null[r]) FlatVector::SetNull(inputs[y_at], phys, true);
			}
			const size_t len = in.x_len ? in.x_len[r] : in.p;
			le[phys].offset = off;
			le[phys].length = len;
			for (size_t j = 0; j < len; ++j) {
				cv[off + j] = j < in.p ? in.x[r * in.p + j] : 0.0;
				if (in.xe_null && j < in.p && in.xe_null[r * in.p + j]) {
					FlatVector::Validity(child).SetInvalid(off + j);
					cv[off + j] = 1e300; // the slot of a NULL holds whatever it holds: the glue must not read it
				}
			}
			off += len;
			if (in.x_null && in.x_null[r]) FlatVector::SetNull(inputs[x_at], phys, true);
			if (w_at != SIZE_MAX) {
				FlatVector::GetData<double>(inputs[w_at])[phys] = in.w ? in.w[r] : 1.0;
				if (in.w_null && in.w_null[r]) FlatVector::SetNull(inputs[w_at], phys, true);
			}
			if (s_at != SIZE_MAX) {
				const uint8_t code = split ? split[r] : 1;
				if (code >= kSplitStringCount) throw std::runtime_error("bad split code");
				if (!kSplitStrings[code]) FlatVector::SetNull(inputs[s_at], phys, true);
				else FlatVector::GetData<string_t>(inputs[s_at])[phys] = inputs[s_at].AddString(kSplitStrings[code]);
			}
		}
		ListVector::SetListSize(inputs[x_at], total);
		if (dictionary)
			for (size_t k = 0; k < data_inputs; ++k) inputs[k].MakeDictionary(sel);
		if (inputs.size() > data_inputs) inputs.back().MakeConstant();
		Vector sv = PointerVector(states.data(), cnt);
		fn_->update(inputs.data(), aid, inputs.size(), sv, cnt);
	}
	FamilyOut FinalizeStates(AggregateInputData &aid, std::vector<data_ptr_t> &states, size_t vector_size) {
		const size_t n = states.size();
		Vector result(fn_->return_type, n ? n : 1);
		for (size_t c0 = 0; c0 < n; c0 += vector_size) {
			const size_t cnt = std::min(vector_size, n - c0);
			Vector sv = PointerVector(states.data() + c0, cnt);
			fn_->finalize(sv, aid, result, cnt, c0);
		}
		FamilyOut out;
		out.is_null.assign(n, 0);
		out.offsets.assign(n + 1, 0);
		for (size_t r = 0; r < n; ++r) {
			const bool valid = FlatVector::Validity(result).RowIsValid(r);
			out.is_null[r] = !valid;
			out.offsets[r + 1] = out.offsets[r];
			if (kind_ == Kind::WINDOW) {
				auto &f = StructVector::GetEntries(result);
				for (int k = 0; k < 3; ++k) out.vals.push_back(valid ? FlatVector::GetData<double>(*f[k])[r] : NAN);
				continue;
			}
			if (!valid) continue;
			const list_entry_t e = ListVector::GetData(result)[r];
			if (e.offset + e.length > ListVector::GetListSize(result)) throw std::runtime_error("finalize wrote a bad LIST entry");
			out.offsets[r + 1] += (int64_t)e.length;
			Vector &child = ListVector::GetEntry(result);
			if (kind_ == Kind::VIF) {
				for (size_t k = 0; k < e.length; ++k) out.vals.push_back(FlatVector::GetData<double>(child)[e.offset + k]);
				continue;
			}
			auto &f = StructVector::GetEntries(child);
			for (size_t k = 0; k < e.length; ++k) {
				const size_t at = e.offset + k;
				uint8_t fl = 0;
				for (int c = 0; c < 4; ++c) {
					const bool ok = FlatVector::Validity(*f[c]).RowIsValid(at);
					if (!ok) fl |= (uint8_t)(1u << c);
					out.vals.push_back(ok ? FlatVector::GetData<double>(*f[c])[at] : NAN);
				}
				if (FlatVector::GetData<bool>(*f[4])[at]) fl |= 16;
				out.flags.push_back(fl);
			}
		}
		return out;
	}
	static void Append(FamilyOut &all, const FamilyOut &part) {
		all.is_null.insert(all.is_null.end(), part.is_null.begin(), part.is_null.end());
		all.vals.insert(all.vals.end(), part.vals.begin(), part.vals.end());
	}

	ExtensionLoader loader_;
	ClientContext context_;
	std::unique_ptr<AggregateFunction> fn_;
	unique_ptr<FunctionData> bind_;
	Kind kind_ = Kind::PREDICT_AGG;
	bool weighted_ = false, with_split_ = false;
};

} // namespace glue_driver
