// sharded_arena.hpp — hash-partitioned ingest across the GPUs of one node (BASELINE north_star: "groups are
// hash-partitioned across the 8 GPUs"): W device states, one per GPU, rows routed by
//     shard = hash64(group key) % W
// into per-shard page-locked chunk buffers, so that all rows of a key meet on one GPU and no arithmetic ever crosses
// devices (the reference fits groups independently, ols_aggregate.cpp:257-337).  Every shard is an AggArena
// (agg_arena.hpp) with its own context, stream and row log on its own device; shipping to different shards takes
// different locks and uses different PCIe links, so W GPUs ingest W times one GPU's rate.  Finalize = one batched fit
// per device; the records are collected on the host per key.  (One process driving W GPUs is how a DuckDB process
// would use a node; with one process PER GPU the exchange step is anofox_hip_gather_records_device / distributed.py.)
//
// The unit routed is the GROUP KEY (a hash aggregate's grouping value, 64 bits), which callers that see keys — an ETL
// host, the Python mirror, bench tools — pass directly.  DuckDB's aggregate callbacks see state pointers, not keys:
// the glue (fit_agg_hip.cpp) keeps using one AggArena per query; routing its slot numbers through this class works the
// same way (key = slot) once cross-device Combine is added — not built.
#pragma once
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "agg_arena.hpp"

namespace anofox_shim {

// splitmix64 finaliser: the routing hash (mirrored by anofox-statistics_amd/distributed.py::hash64)
inline uint64_t hash64(uint64_t z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
inline uint32_t shard_of_key(uint64_t key, uint32_t n_shards) { return (uint32_t)(hash64(key) % n_shards); }

class ShardedAggArena {
public:
	// devices: GPU of every shard (nullptr = all on the calling thread's current device: the single-GPU rehearsal)
	ShardedAggArena(AnofoxHipBatchOptions options, uint32_t n_shards, const int *devices = nullptr, size_t chunk_rows = (size_t)1 << 18,
	                size_t retain_bytes = (size_t)64 << 30, size_t retain_host_bytes = (size_t)32 << 30)
	    : opt_(options) {
		if (n_shards == 0) n_shards = 1;
		for (uint32_t s = 0; s < n_shards; ++s) {
			auto sh = std::make_unique<Shard>();
			sh->arena = std::make_unique<AggArena>(options, chunk_rows, retain_bytes / n_shards, retain_host_bytes / n_shards, devices ? devices[s] : -1);
			shards_.push_back(std::move(sh));
		}
	}
	uint32_t ShardCount() const { return (uint32_t)shards_.size(); }
	size_t FeatureCount() const {
		for (auto &sh : shards_)
			if (sh->arena->FeatureCount()) return sh->arena->FeatureCount();
		return 0;
	}
	uint64_t RowsAccepted() const {
		uint64_t n = 0;
		for (auto &sh : shards_) n += sh->arena->RowsAccepted();
		return n;
	}
	uint64_t RowsOfShard(uint32_t s) const { return shards_[s]->arena->RowsAccepted(); }
	size_t KeysOfShard(uint32_t s) const {
		std::lock_guard<std::mutex> lk(shards_[s]->mu);
		return shards_[s]->keys.size();
	}
	int64_t Unrefined() const {
		int64_t n = 0;
		for (auto &sh : shards_) n += sh->arena->Unrefined();
		return n;
	}

	// One Update call of one thread: rows with their group keys, in arrival order.
	class Writer {
	public:
		explicit Writer(ShardedAggArena &a) : a_(a), w_(a.shards_.size()) {}
		// registers the key (a group exists even if every row of it is skipped) without a row
		void Touch(uint64_t key) { (void)SlotOf(key, shard_of_key(key, (uint32_t)w_.size())); }
		double *Begin(uint64_t key, double y, size_t n_features, double w = 1.0) {
			const uint32_t s = shard_of_key(key, (uint32_t)w_.size());
			const uint32_t slot = SlotOf(key, s);
			if (!w_[s]) w_[s] = std::make_unique<AggArena::Writer>(*a_.shards_[s]->arena);
			return w_[s]->Begin(slot, y, n_features, w);
		}
		void Append(uint64_t key, double y, const double *x, size_t n_features, double w = 1.0) {
			memcpy(Begin(key, y, n_features, w), x, n_features * sizeof(double));
		}

	private:
		uint32_t SlotOf(uint64_t key, uint32_t s) {
			auto hit = cache_.find(key);
			if (hit != cache_.end()) return hit->second;
			Shard &sh = *a_.shards_[s];
			uint32_t slot;
			{
				std::lock_guard<std::mutex> lk(sh.mu);
				auto it = sh.slot_of.find(key);
				if (it == sh.slot_of.end()) {
					slot = sh.arena->NewSlot();
					sh.slot_of.emplace(key, slot);
					sh.keys.push_back(key);
					sh.slots.push_back(slot);
				} else {
					slot = it->second;
				}
			}
			cache_.emplace(key, slot);
			return slot;
		}
		ShardedAggArena &a_;
		std::vector<std::unique_ptr<AggArena::Writer>> w_;
		std::unordered_map<uint64_t, uint32_t> cache_; // keys this call has seen: no lock for their later rows
	};

	// Finalize: every key of every shard (shard 0's keys in first-seen order, then shard 1's, ...) with its records.
	// The shards are fitted concurrently, one host thread each (each is one batched call on its own device).
	void Fetch(std::vector<uint64_t> &keys, std::vector<double> &core, std::vector<double> *inf, std::vector<int> &status) {
		const size_t W = shards_.size();
		std::vector<size_t> base(W + 1, 0);
		for (size_t s = 0; s < W; ++s) {
			std::lock_guard<std::mutex> lk(shards_[s]->mu);
			base[s + 1] = base[s] + shards_[s]->keys.size();
		}
		// the width is the query's: a shard that never saw a row reports status 100 for its (rowless) keys
		const size_t p = FeatureCount(), lc = p + 6, li = 5 * p + 2;
		keys.assign(base[W], 0);
		core.assign(base[W] * lc, 0.0);
		if (inf) inf->assign(opt_.compute_inference ? base[W] * li : 0, 0.0);
		status.assign(base[W], ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS);
		std::vector<std::string> errors(W);
		std::vector<std::thread> th;
		for (size_t s = 0; s < W; ++s)
			th.emplace_back([&, s] {
				try {
					Shard &sh = *shards_[s];
					std::vector<uint32_t> slots;
					{
						std::lock_guard<std::mutex> lk(sh.mu);
						slots.assign(sh.slots.begin(), sh.slots.begin() + (base[s + 1] - base[s]));
						std::copy(sh.keys.begin(), sh.keys.begin() + (base[s + 1] - base[s]), keys.begin() + base[s]);
					}
					if (slots.empty()) return;
					const bool has_rows = sh.arena->FeatureCount() != 0;
					sh.arena->Fetch(slots.data(), slots.size(), has_rows ? core.data() + base[s] * lc : nullptr,
					                (has_rows && inf && !inf->empty()) ? inf->data() + base[s] * li : nullptr, status.data() + base[s]);
				} catch (const std::exception &e) {
					errors[s] = e.what();
				}
			});
		for (auto &t : th) t.join();
		for (auto &e : errors)
			if (!e.empty()) throw std::runtime_error(e);
	}

private:
	struct Shard {
		std::unique_ptr<AggArena> arena;
		mutable std::mutex mu; // the key table
		std::unordered_map<uint64_t, uint32_t> slot_of;
		std::vector<uint64_t> keys;  // first-seen order
		std::vector<uint32_t> slots; // slot of keys[k]
	};
	AnofoxHipBatchOptions opt_;
	std::vector<std::unique_ptr<Shard>> shards_;
};

} // namespace anofox_shim
