// ingest.hip — streaming ingest: the device half of the aggregates' Update / Combine.
//
// Replaces the reference's per-group row buffers (src/aggregate_functions/ols_aggregate.cpp:19-42), the per-row
// scatter of Update (:120-186, p + 1 push_backs into random groups) and the O(n p) buffer copies of Combine
// (:189-234) — and ridge_aggregate.cpp:124-240, wls_aggregate.cpp:122-253 likewise.  Rows never become resident:
// a chunk of rows in ARRIVAL order (state index per row, y, row-major x as in DuckDB's LIST child, optional w) is
// folded into one O(p^2) moment record per state — the same record (common.h MomentLayout<P>) that the batch path's
// accumulate kernel writes, so Finalize is the unchanged solve kernel.
//
// One pass over a chunk (<= kIngestChunkRows rows), nothing synchronises with the host:
//   keys    key = state index, or n_slots for rows the aggregate skips;
//   sort    stable radix sort of (key, row number) on the bits that n_slots needs ((r4) radix_sort.h: hand-written,
//           8-bit digits, three launches per digit; rounds 2-3 called rocPRIM here): the rows of a state become one
//           RUN, still in arrival order, so "the first valid row" (ols.rs:76-87) is the one the reference's buffer
//           would hold first;
//   bounds  run boundaries -> run_start / run_end per state and a list of the states present in the chunk
//           (one atomic per WORKGROUP: (r4) one per wavefront was 16 384 atomics on one address per 2^20 rows —
//           0.19 ms of a 0.96 ms chunk);
//   runs    one wavefront per run, TRANSPOSED against the batch kernel: lane k owns moment k (s_a, q_ab or sw)
//           and every lane walks the run's rows, so there is no cross-lane reduction and the state record is
//           read, updated and written back in place, one coalesced 528-byte access each way at p = 8.  A state
//           that already holds rows keeps its shift (its first valid row ever), so the update is a plain sum;
//           an empty state takes the run's first valid row.  No atomics on the moments: a state appears in one
//           run per chunk and chunks are ordered by the stream — results do not depend on timing.
//   pieces  a run of more than kIngestPieceRows rows (few huge groups) is cut into pieces, one wavefront each,
//           into scratch records; the wavefront that finishes a run's last piece merges them into the state
//           with the shift-moving identity  q_ij += q_b,ij + d_i s_b,j + d_j s_b,i + sw_b d_i d_j.
//
// Bound: with rows arriving in random state order every row costs a read-modify-write of a 528-byte record —
// ~1.1 KB of HBM traffic per 72-byte row — which at 4 TB/s is still 4-5x the 55 GB/s at which PCIe delivers
// rows; sorted arrival is a streaming read.  The ingest is PCIe-bound from host memory.
#include "common.h"
#include "radix_sort.h"

namespace anofox {

namespace {

struct PieceHeader {
	int32_t piece_total;
	int32_t big_total;
	int32_t pad[14];
};
struct PieceBig {
	uint32_t slot;
	int32_t base;   // first piece
	int32_t npiece;
	int32_t done;
};
struct PieceEntry {
	int64_t lo, hi; // positions in the sorted order
	int32_t big;
	int32_t pad;
};
__host__ __device__ inline PieceHeader *pt_header(void *t) { return static_cast<PieceHeader *>(t); }
__host__ __device__ inline PieceBig *pt_big(void *t) { return reinterpret_cast<PieceBig *>(pt_header(t) + 1); }
__host__ __device__ inline PieceEntry *pt_entries(void *t) { return reinterpret_cast<PieceEntry *>(pt_big(t) + kIngestMaxBig); }
__host__ __device__ inline double *pt_records(void *t) { return reinterpret_cast<double *>(pt_entries(t) + kIngestMaxPieces); }

__global__ __launch_bounds__(256) void ingest_keys_kernel(IngestArgs a) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n) return;
	const uint32_t s = a.slot[i];
	bool ok = !a.valid || a.valid[i] != 0;
	if (ok && (int64_t)s >= a.n_slots) {
		ok = false;
		a.counters[1] = 1; // sticky error flag (every writer stores the same value)
	}
	a.keys_in[i] = ok ? s : (uint32_t)a.n_slots;
}

__global__ __launch_bounds__(256) void ingest_bounds_kernel(IngestArgs a) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & 63;
	bool start = false;
	uint32_t k = 0;
	if (i < a.n) {
		k = a.keys_out[i];
		if ((int64_t)k < a.n_slots) {
			start = i == 0 || a.keys_out[i - 1] != k;
			const bool end = i == a.n - 1 || a.keys_out[i + 1] != k;
			if (start) a.run_start[k] = (int32_t)i;
			if (end) a.run_end[k] = (int32_t)(i + 1);
		}
	}
	// append the runs to the chunk's list: the wavefronts' counts meet in LDS, one atomic per workgroup
	__shared__ int wave_count[4];
	__shared__ int block_base;
	const int w = threadIdx.x >> 6;
	const unsigned long long b = __ballot(start);
	if (lane == 0) wave_count[w] = (int)__popcll(b);
	__syncthreads();
	if (threadIdx.x == 0) {
		const int total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
		block_base = total ? atomicAdd(&a.counters[0], total) : 0;
	}
	__syncthreads();
	if (!start) return;
	int base = block_base;
	for (int k2 = 0; k2 < w; ++k2) base += wave_count[k2];
	a.run_list[base + (int)__popcll(b & ((1ull << lane) - 1ull))] = k;
}

// this lane's moment of the record: s_i (lane < Z), q_ij (upper triangle, row-major) or sw
template <int P>
__device__ __forceinline__ void lane_moment(int lane, int &mi, int &mj, bool &is_s, bool &is_q, bool &is_sw) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	mi = mj = 0;
	is_s = lane < Z;
	is_q = lane >= L::OFF_Q && lane < L::OFF_SW;
	is_sw = lane == L::OFF_SW;
	if (is_s) mi = mj = lane;
	if (is_q) {
		int k = lane - L::OFF_Q;
#pragma unroll
		for (int a = 0; a < Z; ++a) {
			if (k >= 0 && k < Z - a) { mi = a; mj = a + k; k = -1; }
			else if (k >= 0) k -= Z - a;
		}
	}
}

// Fold the rows at sorted positions [lo, hi) into `rec`.  from_state: `rec` is the state's record and is updated in
// place; otherwise `rec` is a scratch record that starts empty.
template <int P, bool WEIGHTED, bool CENTER>
__device__ __forceinline__ void ingest_rows(const IngestArgs &a, int64_t lo, int64_t hi, double *rec, bool from_state, int lane) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	int mi, mj;
	bool is_s, is_q, is_sw;
	lane_moment<P>(lane, mi, mj, is_s, is_q, is_sw);

	double acc = 0.0, fa = 0.0, fb = 0.0, cnt = 0.0;
	unsigned mask = 0;
	bool have = false;
	if (from_state) {
		// (r4) every field is loaded at once and SELECTED on the row count: with the loads behind `if (count > 0)` a run paid two
		// dependent round trips to its record before its first row (an empty state's record is all zeros)
		const double c0 = rec[L::OFF_CNT], m0 = rec[L::OFF_MASK];
		const double f_a = rec[L::OFF_FIRST + mi], f_b = rec[L::OFF_FIRST + mj];
		const double a0 = rec[lane < L::KRED ? lane : 0];
		have = c0 > 0.0; // wave-uniform
		cnt = have ? c0 : 0.0;
		mask = have ? (unsigned)m0 : 0u;
		fa = have ? f_a : 0.0;
		fb = have ? f_b : 0.0;
		acc = have && lane < L::KRED ? a0 : 0.0;
	}
	// operand a / b of this lane: column mi / mj of the row (x is row-major, y is its own array)
	const double *pa = mi < P ? a.x + mi : a.y;
	const double *pb = mj < P ? a.x + mj : a.y;
	const int64_t sa = mi < P ? P : 1, sb = mj < P ? P : 1;

	constexpr int U = 4; // rows in flight
	for (int64_t i = lo; i < hi; i += U) {
		double za[U], zb[U], ww[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t pos = i + u < hi ? i + u : hi - 1;
			const int64_t r = (int64_t)a.rows_out[pos];
			za[u] = pa[r * sa];
			zb[u] = pb[r * sb];
			ww[u] = WEIGHTED ? a.w[r] : 1.0;
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			if (i + u >= hi) break; // wave-uniform
			// row filter: everything finite (and w > 0), ols.rs:59-66 / wls.rs:76-86; lanes 0..Z-1 cover every column
			bool bad = !(isfinite(za[u]) && isfinite(zb[u]));
			if (WEIGHTED) bad = bad || !(ww[u] > 0.0) || !isfinite(ww[u]);
			if (__ballot(bad) != 0ull) continue;
			if (!have) {
				fa = za[u];
				fb = zb[u];
				have = true;
			}
			cnt += 1.0;
			// constant-column test against the first valid row (ols.rs:76-87)
			mask |= (unsigned)__ballot(lane < P && !(fabs(za[u] - fa) < 1e-10));
			const double da = CENTER ? za[u] - fa : za[u];
			const double db = CENTER ? zb[u] - fb : zb[u];
			const double wd = WEIGHTED ? ww[u] * da : da;
			if (is_q) acc = fma(wd, db, acc);
			else if (is_s) acc += wd;
			else if (is_sw) acc += ww[u];
		}
	}
	if (lane < L::KRED) rec[lane] = acc;
	double e = fa; // lanes < Z own s_lane, so fa = first[lane]
	e = (lane == Z) ? cnt : e;
	e = (lane == Z + 1) ? (double)(mask & ((1u << P) - 1u)) : e;
	if (lane < Z + 2) rec[L::KRED + lane] = e;
}

// state += piece records (in order).  A piece shifted by ITS first valid row is moved to the state's shift:
// with delta = first_b - first_state:  s += s_b + sw_b delta,  q_ij += q_b,ij + delta_i s_b,j + delta_j s_b,i + sw_b delta_i delta_j.
template <int P, bool CENTER>
__device__ __forceinline__ void merge_into_state(double *state, const double *pieces, int npiece, int lane) {
	using L = MomentLayout<P>;
	constexpr int Z = L::Z;
	int mi, mj;
	bool is_s, is_q, is_sw;
	lane_moment<P>(lane, mi, mj, is_s, is_q, is_sw);
	double total = 0.0, cnt = 0.0, ai = 0.0, aj = 0.0, a_first = 0.0;
	unsigned mask = 0;
	bool have = false;
	{
		const double c0 = state[L::OFF_CNT];
		if (c0 > 0.0) {
			have = true;
			cnt = c0;
			mask = (unsigned)state[L::OFF_MASK];
			ai = state[L::OFF_FIRST + mi];
			aj = state[L::OFF_FIRST + mj];
			a_first = lane < Z ? state[L::OFF_FIRST + lane] : 0.0;
			total = lane < L::KRED ? state[lane] : 0.0;
		}
	}
	for (int t = 0; t < npiece; ++t) {
		const double *rb = pieces + (int64_t)t * L::REC;
		const double cb = rb[L::OFF_CNT];
		if (!(cb > 0.0)) continue; // wave-uniform
		if (!have) {
			ai = rb[L::OFF_FIRST + mi];
			aj = rb[L::OFF_FIRST + mj];
			a_first = lane < Z ? rb[L::OFF_FIRST + lane] : 0.0;
			have = true;
		}
		const double swb = rb[L::OFF_SW];
		const double di = CENTER ? rb[L::OFF_FIRST + mi] - ai : 0.0;
		const double dj = CENTER ? rb[L::OFF_FIRST + mj] - aj : 0.0;
		double v = 0.0;
		if (is_s) v = rb[L::OFF_S + mi] + swb * di;
		else if (is_q) v = rb[lane] + di * rb[L::OFF_S + mj] + dj * rb[L::OFF_S + mi] + swb * di * dj;
		else if (is_sw) v = swb;
		total += v;
		cnt += cb;
		unsigned m = (unsigned)rb[L::OFF_MASK];
		const bool moved = lane < P && !(fabs(rb[L::OFF_FIRST + lane] - a_first) < 1e-10);
		m |= (unsigned)__ballot(moved);
		mask |= m;
	}
	if (lane < L::KRED) state[lane] = total;
	double e = a_first;
	e = (lane == Z) ? cnt : e;
	e = (lane == Z + 1) ? (double)(mask & ((1u << P) - 1u)) : e;
	if (lane < Z + 2) state[L::KRED + lane] = e;
}

template <int P, bool WEIGHTED, bool CENTER>
__device__ __forceinline__ void ingest_one_run(const IngestArgs &a, uint32_t slot, int64_t lo, int64_t hi, int lane) {
	using L = MomentLayout<P>;
	{
		if (lane == 0) a.n_accum[slot] += hi - lo; // rows Update accepted, valid or not (ols_aggregate.cpp:176)
		if (hi - lo > kIngestPieceRows) {
			// a single wavefront walks ~10 rows per microsecond: hand a long run to ingest_pieces_kernel in pieces
			PieceHeader *h = pt_header(a.piece_table);
			const int np = (int)((hi - lo + kIngestPieceRows - 1) / kIngestPieceRows);
			int big = -1, base = -1;
			if (lane == 0) {
				base = reserve_table_entries(&h->piece_total, np, kIngestMaxPieces);
				if (base >= 0) big = reserve_table_entries(&h->big_total, 1, kIngestMaxBig);
			}
			big = __builtin_amdgcn_readfirstlane(big);
			base = __builtin_amdgcn_readfirstlane(base);
			if (base >= 0) {
				if (big >= 0 && lane == 0) {
					PieceBig b;
					b.slot = slot; b.base = base; b.npiece = np; b.done = 0;
					pt_big(a.piece_table)[big] = b;
				}
				for (int k = lane; k < np; k += 64) {
					PieceEntry e;
					e.lo = lo + (int64_t)k * kIngestPieceRows;
					e.hi = e.lo + kIngestPieceRows < hi ? e.lo + kIngestPieceRows : hi;
					e.big = big; e.pad = 0;
					if (big < 0) e.hi = e.lo; // reserved without a run slot: empty, unclaimed
					pt_entries(a.piece_table)[base + k] = e;
				}
				if (big >= 0) return;
			}
		}
		ingest_rows<P, WEIGHTED, CENTER>(a, lo, hi, a.moments + (int64_t)slot * L::REC, true, lane);
	}
}

template <int P, bool WEIGHTED, bool CENTER>
__global__ __launch_bounds__(256) void ingest_runs_kernel(IngestArgs a) {
	const int lane = threadIdx.x & 63;
	const int n_runs = a.counters[0];
	const int n_waves = (int)gridDim.x * 4;
	// (r4) the run list is walked two runs ahead: slot number of run v + 2 and bounds of run v + 1 are loaded before run v is
	// folded, so that a run's chain of dependent loads is record / row numbers -> rows, not list -> bounds -> record -> ... -> rows
	// (a run of a few rows — rows in random state order — is latency, not bytes)
	int v = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (v >= n_runs) return;
	uint32_t slot = a.run_list[v];
	uint32_t slot_1 = v + n_waves < n_runs ? a.run_list[v + n_waves] : slot;
	int64_t lo = a.run_start[slot], hi = a.run_end[slot];
	for (; v < n_runs; v += n_waves) {
		const int v2 = v + 2 * n_waves;
		const uint32_t slot_2 = v2 < n_runs ? a.run_list[v2] : slot_1;
		const int64_t lo_1 = a.run_start[slot_1], hi_1 = a.run_end[slot_1];
		ingest_one_run<P, WEIGHTED, CENTER>(a, slot, lo, hi, lane);
		slot = slot_1;
		lo = lo_1;
		hi = hi_1;
		slot_1 = slot_2;
	}
}

template <int P, bool WEIGHTED, bool CENTER>
__global__ __launch_bounds__(256) void ingest_pieces_kernel(IngestArgs a) {
	using L = MomentLayout<P>;
	const int lane = threadIdx.x & 63;
	const int v = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int)blockIdx.x * 4;
	PieceHeader *h = pt_header(a.piece_table);
	if (v >= h->piece_total) return;
	const PieceEntry e = pt_entries(a.piece_table)[v];
	if (e.big < 0) return;
	double *recs = pt_records(a.piece_table);
	ingest_rows<P, WEIGHTED, CENTER>(a, e.lo, e.hi, recs + (int64_t)v * L::REC, false, lane);
	__threadfence(); // this piece's record before the counter
	PieceBig *b = pt_big(a.piece_table) + e.big;
	int old = 0;
	if (lane == 0) old = atomicAdd(&b->done, 1);
	old = __builtin_amdgcn_readfirstlane(old);
	if (old != b->npiece - 1) return;
	__threadfence(); // every other piece's record after the counter
	merge_into_state<P, CENTER>(a.moments + (int64_t)b->slot * L::REC, recs + (int64_t)b->base * L::REC, b->npiece, lane);
}

template <int P, bool CENTER>
__global__ __launch_bounds__(256) void ingest_combine_kernel(double *moments, int64_t *n_accum, int64_t n_slots, const uint32_t *src,
                                                             const uint32_t *dst, int64_t n_pairs, int preserve) {
	using L = MomentLayout<P>;
	const int lane = threadIdx.x & 63;
	const int64_t v = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (int64_t)blockIdx.x * 4;
	if (v >= n_pairs) return;
	const uint32_t s = src[v], d = dst[v];
	if ((int64_t)s >= n_slots || (int64_t)d >= n_slots || s == d) return;
	double *srec = moments + (int64_t)s * L::REC;
	merge_into_state<P, CENTER>(moments + (int64_t)d * L::REC, srec, 1, lane);
	if (!preserve)
		for (int k = lane; k < L::REC; k += 64) srec[k] = 0.0; // the source state is reset (Combine moves its rows)
	if (lane == 0) {
		n_accum[d] += n_accum[s];
		if (!preserve) n_accum[s] = 0;
	}
}

template <int P>
hipError_t launch_chunk_p(const IngestArgs &a, hipStream_t st) {
	const unsigned rows_grid = (unsigned)((a.n + 255) / 256);
	hipLaunchKernelGGL(ingest_keys_kernel, dim3(rows_grid), dim3(256), 0, st, a);
	// stable sort of (state index, row number) on the bits that n_slots (the key of skipped rows) needs
	unsigned end_bit = 1;
	while (end_bit < 32 && ((uint64_t)a.n_slots >> end_bit) != 0) ++end_bit;
	hipError_t rc = rsort::sort<uint32_t, true>(a.keys_in, a.keys_out, a.rows_out, (size_t)a.n, end_bit, a.sort_temp, a.sort_temp_bytes, st);
	if (rc != hipSuccess) return rc;
	hipLaunchKernelGGL(ingest_bounds_kernel, dim3(rows_grid), dim3(256), 0, st, a);
	unsigned run_grid = (unsigned)((a.n + 3) / 4);
	if (run_grid > 16384u) run_grid = 16384u;
	const dim3 piece_grid((unsigned)((kIngestMaxPieces + 3) / 4)); // idle unless a run exceeded kIngestPieceRows
#define ANOFOX_INGEST_LAUNCH(W, C)                                                                  \
	do {                                                                                            \
		hipLaunchKernelGGL((ingest_runs_kernel<P, W, C>), dim3(run_grid), dim3(256), 0, st, a);     \
		hipLaunchKernelGGL((ingest_pieces_kernel<P, W, C>), piece_grid, dim3(256), 0, st, a);       \
	} while (0)
	if (a.weighted) {
		if (a.center) ANOFOX_INGEST_LAUNCH(true, true);
		else ANOFOX_INGEST_LAUNCH(true, false);
	} else {
		if (a.center) ANOFOX_INGEST_LAUNCH(false, true);
		else ANOFOX_INGEST_LAUNCH(false, false);
	}
#undef ANOFOX_INGEST_LAUNCH
	return hipGetLastError();
}

template <int P>
hipError_t launch_combine_p(double *moments, int64_t *n_accum, int64_t n_slots, const uint32_t *src, const uint32_t *dst,
                            int64_t n_pairs, int center, int preserve, hipStream_t st) {
	const dim3 grid((unsigned)((n_pairs + 3) / 4)), block(256);
	if (center) hipLaunchKernelGGL((ingest_combine_kernel<P, true>), grid, block, 0, st, moments, n_accum, n_slots, src, dst, n_pairs, preserve);
	else hipLaunchKernelGGL((ingest_combine_kernel<P, false>), grid, block, 0, st, moments, n_accum, n_slots, src, dst, n_pairs, preserve);
	return hipGetLastError();
}

} // namespace

size_t ingest_piece_table_bytes(int p) {
	return sizeof(PieceHeader) + sizeof(PieceBig) * kIngestMaxBig + sizeof(PieceEntry) * kIngestMaxPieces +
	       sizeof(double) * (size_t)kIngestMaxPieces * (size_t)moment_record_len(p);
}

size_t ingest_sort_temp_bytes(int64_t n) {
	return rsort::temp_bytes<uint32_t, true>((size_t)n);
}

hipError_t launch_ingest_chunk(const IngestArgs &a, hipStream_t stream) {
	if (a.n <= 0) return hipSuccess;
	if (a.n > kIngestChunkRows || a.n_slots <= 0 || a.n_slots > (int64_t)0x7fffffff) return hipErrorInvalidValue;
	switch (a.p) {
	case 1: return launch_chunk_p<1>(a, stream);
	case 2: return launch_chunk_p<2>(a, stream);
	case 3: return launch_chunk_p<3>(a, stream);
	case 4: return launch_chunk_p<4>(a, stream);
	case 5: return launch_chunk_p<5>(a, stream);
	case 6: return launch_chunk_p<6>(a, stream);
	case 7: return launch_chunk_p<7>(a, stream);
	case 8: return launch_chunk_p<8>(a, stream);
	default: return hipErrorInvalidValue;
	}
}

namespace {
// records and accepted-row counts of the listed slots -> contiguous rows k = 0 .. n_list - 1 (subset Finalize)
__global__ void ingest_gather_slots_kernel(const double *moments, const int64_t *n_accum, const uint32_t *list, int64_t n_list, int rec,
                                           double *out_m, int64_t *out_n) {
	for (int64_t k = blockIdx.x; k < n_list; k += gridDim.x) {
		const int64_t s = list[k];
		for (int j = threadIdx.x; j < rec; j += blockDim.x) out_m[k * rec + j] = moments[s * rec + j];
		if (threadIdx.x == 0) out_n[k] = n_accum[s];
	}
}
// Destroy of aggregate states: the listed slots become empty again (all-zero record, no accepted rows)
__global__ void ingest_clear_slots_kernel(double *moments, int64_t *n_accum, const uint32_t *list, int64_t n_list, int rec) {
	for (int64_t k = blockIdx.x; k < n_list; k += gridDim.x) {
		const int64_t s = list[k];
		for (int j = threadIdx.x; j < rec; j += blockDim.x) moments[s * rec + j] = 0.0;
		if (threadIdx.x == 0) n_accum[s] = 0;
	}
}
} // namespace

hipError_t launch_ingest_gather_slots(const double *moments, const int64_t *n_accum, const uint32_t *list, int64_t n_list, int p, double *out_m,
                                      int64_t *out_n, hipStream_t st) {
	if (n_list <= 0) return hipSuccess;
	const unsigned grid = (unsigned)(n_list < 65535 ? n_list : 65535);
	hipLaunchKernelGGL(ingest_gather_slots_kernel, dim3(grid), dim3(64), 0, st, moments, n_accum, list, n_list, moment_record_len(p), out_m, out_n);
	return hipGetLastError();
}

namespace {
// the inverse of ingest_gather_slots_kernel: contiguous records k = 0 .. n_list - 1 -> the listed slots (cross-device Combine)
__global__ void ingest_scatter_slots_kernel(double *moments, int64_t *n_accum, const uint32_t *list, int64_t n_list, int rec,
                                            const double *in_m, const int64_t *in_n) {
	for (int64_t k = blockIdx.x; k < n_list; k += gridDim.x) {
		const int64_t s = list[k];
		for (int j = threadIdx.x; j < rec; j += blockDim.x) moments[s * rec + j] = in_m[k * rec + j];
		if (threadIdx.x == 0) n_accum[s] = in_n[k];
	}
}
} // namespace

hipError_t launch_ingest_scatter_slots(double *moments, int64_t *n_accum, const uint32_t *list, int64_t n_list, int p, const double *in_m,
                                       const int64_t *in_n, hipStream_t st) {
	if (n_list <= 0) return hipSuccess;
	const unsigned grid = (unsigned)(n_list < 65535 ? n_list : 65535);
	hipLaunchKernelGGL(ingest_scatter_slots_kernel, dim3(grid), dim3(64), 0, st, moments, n_accum, list, n_list, moment_record_len(p), in_m, in_n);
	return hipGetLastError();
}

hipError_t launch_ingest_clear_slots(double *moments, int64_t *n_accum, const uint32_t *list, int64_t n_list, int p, hipStream_t st) {
	if (n_list <= 0) return hipSuccess;
	const unsigned grid = (unsigned)(n_list < 65535 ? n_list : 65535);
	hipLaunchKernelGGL(ingest_clear_slots_kernel, dim3(grid), dim3(64), 0, st, moments, n_accum, list, n_list, moment_record_len(p));
	return hipGetLastError();
}

hipError_t launch_ingest_combine(double *moments, int64_t *n_accum, int64_t n_slots, const uint32_t *src, const uint32_t *dst,
                                 int64_t n_pairs, int p, int center, int preserve, hipStream_t stream) {
	if (n_pairs <= 0) return hipSuccess;
	switch (p) {
	case 1: return launch_combine_p<1>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	case 2: return launch_combine_p<2>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	case 3: return launch_combine_p<3>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	case 4: return launch_combine_p<4>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	case 5: return launch_combine_p<5>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	case 6: return launch_combine_p<6>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	case 7: return launch_combine_p<7>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	case 8: return launch_combine_p<8>(moments, n_accum, n_slots, src, dst, n_pairs, center, preserve, stream);
	default: return hipErrorInvalidValue;
	}
}

} // namespace anofox
