// host_api.hip — host side of libanofox_stats_hip.so: contexts, workspace, the batched entry points and
// the reference-compatible single-group symbols (include/anofox_stats_hip.h).
//
// The single-group symbols keep the conventions of the reference's Rust FFI shims
// (crates/anofox-stats-ffi/src/lib.rs:98-311, 984-1155, 1384-1555): error struct reset first, NULL
// out_core / x / x_count == 0 -> InvalidInput, NULL entries replaced by NaN through the validity bitmask
// (types.rs:66-89), outputs malloc'ed and released only by anofox_free_result_*, nothing handed out on
// failure.  They run the same GPU kernels as the batch path with a batch of one group.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "host_math.h"

using namespace anofox;

#include "context.h"

using namespace anofox::host;

namespace {

struct Workspace {
	void *seg_table; // SegHeader + tables + segment records (common.h)
	double *moments;
	double *refine_vec;
	int32_t *refine_list;
	int32_t *refine_count;
	void *tcrit_table;
};

bool carve_workspace(AnofoxHipContext *ctx, int64_t G, int p, Workspace *out, AnofoxError *e) {
	const size_t rec = (size_t)moment_record_len(p);
	const size_t b_mom = align_up((size_t)G * rec * sizeof(double), 256);
	const size_t b_rss = align_up((size_t)G * (size_t)refine_vec_len((int)p) * sizeof(double), 256);
	const size_t b_lst = align_up((size_t)G * sizeof(int32_t), 256);
	const size_t b_seg = align_up(seg_table_bytes(p), 256);
	const size_t total = b_mom + b_rss + b_lst + 256 + kTcritTableBytes + b_seg;
	if (!ensure_buffer(&ctx->ws, &ctx->ws_bytes, total, "workspace", e)) return false;
	char *base = (char *)ctx->ws;
	out->seg_table = base + b_mom + b_rss + b_lst + 256 + kTcritTableBytes; // its header is zeroed with the counters
	out->moments = (double *)base;
	out->refine_vec = (double *)(base + b_mom);
	out->refine_list = (int32_t *)(base + b_mom + b_rss);
	out->refine_count = (int32_t *)(base + b_mom + b_rss + b_lst);
	out->tcrit_table = base + b_mom + b_rss + b_lst + 256;
	return true;
}

// Wide designs (8 < p <= 128): the same four stages on the MFMA / LDS kernels, slab by slab.
bool run_wide_batch(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const int64_t *d_off,
                    const double *d_y, const double *const *x_cols, const double *d_w,
                    const AnofoxHipBatchOptions &opt, double *d_core, double *d_inf, const int64_t *d_rule_counts,
                    AnofoxError *e) {
	const int T = wide_tiles((int)p);
	const size_t rec_bytes = (size_t)wide_record_len(T) * sizeof(double);
	// the moment scratch is reused by slabs of groups: at most ~1 GiB of it is live
	int64_t slab = (int64_t)((size_t)1 << 30) / (int64_t)rec_bytes;
	if (slab < 256) slab = 256;
	if (slab > G) slab = G;
	const size_t b_mom = align_up((size_t)slab * rec_bytes, 256);
	const size_t b_rss = align_up((size_t)G * (size_t)refine_vec_len((int)p) * sizeof(double), 256);
	const size_t b_lst = align_up((size_t)slab * sizeof(int32_t), 256);
	const bool mid = solve_mid_supports((int)p);
	static const bool mid_acc_on = !(getenv("ANOFOX_MID_ACC") && atoi(getenv("ANOFOX_MID_ACC")) == 0); // A/B switch for measurements
	const bool mid_acc = mid_acc_on && accumulate_mid_supports((int)p);
	// Several slabs: slab k's solve / refinement kernels run on a second stream while the main stream accumulates slab k + 1
	// into the other moment buffer.  Measured (profiles/r03_slab_overlap.txt): the time between the accumulate kernels
	// drops (cfg5: 5.05 -> 1.66 ms per step) but the accumulate kernels stretch by almost as much (69.7 -> 72.3 ms) — the
	// one-wavefront-per-group solve keeps the CUs busy, it is not idle time to hide: 74.7 -> 73.9 ms per cfg5 step, nothing
	// at n = 1000.  ANOFOX_WIDE_OVERLAP=0: everything on the one stream, as for a single slab.
	static const bool overlap_on = !(getenv("ANOFOX_WIDE_OVERLAP") && atoi(getenv("ANOFOX_WIDE_OVERLAP")) == 0);
	const bool overlap = overlap_on && G > slab;
	const int n_buf = overlap ? 2 : 1;
	// very large groups are split into row segments (accumulate_mid.hip: a wave each; accumulate_wide.hip: a workgroup each)
	const size_t b_seg = align_up(mid_acc ? wide_seg_table_bytes(T, kSegMaxBig, kSegMaxSegments)
	                                      : wide_seg_table_bytes(T, kWideSegMaxBig, kWideSegMaxSegments), 256);
	// workspace: moments x n_buf | refine vectors (all groups) | refine list x n_buf | counters (256 B) x n_buf | t table | segment table
	// (the segment table per moment buffer: slab k's segment kernel runs on the solve stream while slab k + 1 registers its groups)
	if (!ensure_buffer(&ctx->ws, &ctx->ws_bytes, n_buf * (b_mom + b_lst + 256 + b_seg) + b_rss + kTcritTableBytes, "workspace", e)) return false;
	char *base = (char *)ctx->ws;
	char *const w_rss = base + n_buf * b_mom, *const w_lst = w_rss + b_rss, *const w_cnt = w_lst + n_buf * b_lst;
	char *const w_tcrit = w_cnt + n_buf * 256, *const w_seg = w_tcrit + kTcritTableBytes;

	WideArgs a;
	memset(&a, 0, sizeof a);
	a.row_offsets = d_off;
	a.y = d_y;
	for (size_t j = 0; j < p; ++j) a.x_table[j] = x_cols[j];
	if (p < (size_t)kWideMaxP) a.x_table[p] = d_y; // the LDS-DMA kernels walk ONE table: features, then y (lds_dma.h)
	a.w = d_w;
	a.p = (int)p;
	a.model = (int)opt.model;
	a.fit_intercept = opt.fit_intercept ? 1 : 0;
	a.compute_inference = opt.compute_inference ? 1 : 0;
	a.lambda_scaling = (int)opt.lambda_scaling;
	a.hc_type = (int)opt.hc_type;
	a.confidence_level = opt.confidence_level;
	a.alpha = opt.alpha;
	a.refine_vec = (double *)w_rss;
	a.tcrit_table = w_tcrit;
	a.core = d_core;
	a.inference = opt.compute_inference ? d_inf : nullptr;
	a.rule_counts = d_rule_counts;
	a.row_ends = ctx->frame_ends;
	static const bool wide_fast_on = !(getenv("ANOFOX_WIDE_FAST") && atoi(getenv("ANOFOX_WIDE_FAST")) == 0); // A/B switch
	a.no_fast_path = wide_fast_on ? 0 : 1;

	hipStream_t st = ctx->stream;
	hipStream_t ss = st; // the stream of the solve / refinement kernels
	if (overlap) {
		if (!ctx->solve_stream && hip_fail(hipStreamCreateWithFlags(&ctx->solve_stream, hipStreamNonBlocking), "hipStreamCreate", e)) return false;
		for (int b = 0; b < 2; ++b) {
			if (!ctx->slab_acc_done[b] && hip_fail(hipEventCreateWithFlags(&ctx->slab_acc_done[b], hipEventDisableTiming), "hipEventCreate", e)) return false;
			if (!ctx->slab_solve_done[b] && hip_fail(hipEventCreateWithFlags(&ctx->slab_solve_done[b], hipEventDisableTiming), "hipEventCreate", e)) return false;
		}
		ss = ctx->solve_stream;
	}
	a.seg_table = w_seg; // (per buffer below)
	a.seg_rows = mid_acc ? seg_rows_for(n_rows) : wide_seg_rows_for(n_rows);
	if (hip_fail(hipMemsetAsync(a.tcrit_table, 0, kTcritTableBytes, st), "hipMemsetAsync", e)) return false;
	// (r4) the slabs' sizes.  With two streams slab k's solve runs under slab k + 1's accumulate kernel, so what is exposed is the LAST
	// slab's solve — a short last slab is welcome — but a last slab much shorter than its predecessor leaves most of the predecessor's
	// solve exposed as well (50 000 x 1000 x 64: 47 527 + 2 473 groups, the solve of the first 1.7 ms, the accumulate kernel of the
	// second 0.4 ms).  A remainder below a quarter of a slab therefore takes groups from the slab before it: 3 : 1.
	std::vector<int64_t> slab_sizes;
	for (int64_t left = G; left > 0;) {
		const int64_t take = left < slab ? left : slab;
		slab_sizes.push_back(take);
		left -= take;
	}
	static const bool slab_balance_on = !(getenv("ANOFOX_WIDE_SLAB_BALANCE") && atoi(getenv("ANOFOX_WIDE_SLAB_BALANCE")) == 0); // A/B switch
	if (const size_t K = slab_sizes.size(); slab_balance_on && K >= 2 && slab_sizes[K - 1] < slab / 4) {
		const int64_t both = slab_sizes[K - 2] + slab_sizes[K - 1];
		slab_sizes[K - 1] = both / 4;
		slab_sizes[K - 2] = both - both / 4;
	}
	int64_t k_slab = 0;
	for (int64_t g0 = 0; g0 < G; g0 += slab_sizes[(size_t)k_slab], ++k_slab) {
		const int buf = overlap ? (int)(k_slab & 1) : 0;
		a.group_base = g0;
		a.n_groups = slab_sizes[(size_t)k_slab];
		a.moments = (double *)(base + buf * b_mom);
		a.refine_list = (int32_t *)(w_lst + buf * b_lst);
		a.refine_count = (int32_t *)(w_cnt + buf * 256);
		a.seg_table = w_seg + buf * b_seg;
		ctx->last_refine_count = a.refine_count;
		// this buffer's previous tenant (slab k - 2) must be through its solve before the buffer is written again
		if (overlap && k_slab >= 2 && hip_fail(hipStreamWaitEvent(st, ctx->slab_solve_done[buf], 0), "hipStreamWaitEvent", e)) return false;
		if (hip_fail(hipMemsetAsync(a.refine_count, 0, 64, st), "hipMemsetAsync", e)) return false; // [0] refine queue, [8] accumulate_wide's redo list
		if (a.seg_table && hip_fail(hipMemsetAsync(a.seg_table, 0, sizeof(SegHeader), st), "hipMemsetAsync", e)) return false;
		hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
		if (ctx->timing) {
			e0 = get_event(ctx);
			e1 = get_event(ctx);
			e2 = get_event(ctx);
			(void)hipEventRecord(e0, st);
		}
		// pipelining gate (anofox_hip_context_set_accumulate_gate): wait before the first accumulate kernel of the
		// call, record after the last one
		if (g0 == 0 && ctx->gate_wait) {
			hipEvent_t gw = ctx->gate_wait;
			ctx->gate_wait = nullptr; // one-shot: the handle belongs to the caller and may be gone after this call
			if (hip_fail(hipStreamWaitEvent(st, gw, 0), "hipStreamWaitEvent", e)) return false;
			if (ctx->timing) (void)hipEventRecord(e0, st);
		}
		// 8 < p <= 32: 4 x 4-block MFMAs (accumulate_quad.hip) or 16 x 16 tiles (accumulate_mid.hip); ANOFOX_MID_QUAD=0/1
		static const bool quad_on = !(getenv("ANOFOX_MID_QUAD") && atoi(getenv("ANOFOX_MID_QUAD")) == 0);
		// (r4: its speculative kernel also takes p = 27 .. 34 of the unweighted fit with an intercept)
		const bool quad = mid_acc_on && quad_on && accumulate_quad_supports((int)p, opt.model == ANOFOX_HIP_MODEL_WLS, opt.fit_intercept, a.no_fast_path != 0);
		// (r4) 34 <= p <= 64: the wave-per-group speculative kernel on LDS-DMA (accumulate_mid.hip) before accumulate_wide's full version
		const bool tile = mid_acc_on && !quad && accumulate_tile_supports((int)p, opt.model == ANOFOX_HIP_MODEL_WLS, opt.fit_intercept, a.no_fast_path != 0);
		// The workgroup-per-group kernel's segment / redo kernels are idle unless a group needs them, yet an idle launch is not
		// free next to another stream's kernel: every one of its workgroups needs a slot (74 KB of LDS, 4 x 237 registers) that
		// the other kernel's wavefronts hold, so its trace duration is the time it QUEUED — 5 us, or ~1 ms behind the previous
		// slab's solve (512 registers per wave, one wave per SIMD): round 3's "bimodal idle launch" (profiles/r04_idle_launches.md).
		// Moving them to the solve stream (ANOFOX_WIDE_SPLIT=1) was built and measured: there the redo kernel's 13 785
		// workgroups queue behind the NEXT slab's accumulate kernel instead (2-18 ms) and take the slab's solve with them — worse;
		// they stay on this stream.
		const bool plain_wide = !quad && !tile && !mid_acc;
		static const bool split_on = getenv("ANOFOX_WIDE_SPLIT") && atoi(getenv("ANOFOX_WIDE_SPLIT")) == 1; // measurement switch
		a.launch_part = (plain_wide && overlap && split_on) ? 1 : 0;
		// (r4) expanding window frames (run_window): every frame is its predecessor plus a row — their records come from
		// accumulate_prefix.hip, n / (2 K) + 1 rows read per frame instead of n / 2.  ANOFOX_FRAMES_PREFIX=0: the kernels below.
		static const bool prefix_on = !(getenv("ANOFOX_FRAMES_PREFIX") && atoi(getenv("ANOFOX_FRAMES_PREFIX")) == 0);
		static const int prefix_k = getenv("ANOFOX_FRAMES_PREFIX_K") ? atoi(getenv("ANOFOX_FRAMES_PREFIX_K")) : 128;
		const bool prefix = prefix_on && ctx->frame_prefix && a.row_ends != nullptr;
		if (hip_fail(prefix ? launch_accumulate_prefix(a, prefix_k > 0 ? prefix_k : 128, st)
		                    : (quad ? launch_accumulate_quad(a, st)
		                            : (tile ? launch_accumulate_tile(a, st) : (mid_acc ? launch_accumulate_mid(a, st) : launch_accumulate_wide(a, st)))),
		             "wide accumulate kernel launch", e))
			return false;
		if (ctx->timing) (void)hipEventRecord(e1, st);
		if (g0 + a.n_groups >= G && ctx->gate_record) {
			hipEvent_t gr = ctx->gate_record;
			ctx->gate_record = nullptr;
			if (hip_fail(hipEventRecord(gr, st), "hipEventRecord", e)) return false;
		}
		if (overlap) {
			if (hip_fail(hipEventRecord(ctx->slab_acc_done[buf], st), "hipEventRecord", e)) return false;
			if (hip_fail(hipStreamWaitEvent(ss, ctx->slab_acc_done[buf], 0), "hipStreamWaitEvent", e)) return false;
			if (a.launch_part == 1) {
				a.launch_part = 2;
				if (hip_fail(launch_accumulate_wide(a, ss), "wide accumulate follow-up launch", e)) return false;
				a.launch_part = 0;
			}
		}
		// moderately wide designs: one lane per group (solve_mid.hip); beyond that one workgroup per group
		auto solve = [&](int mode) { return mid ? launch_solve_mid(a, mode, ss) : launch_solve_wide(a, mode, ss); };
		// the primary solve of every width runs with one wavefront per group and the matrix in registers (solve_tiles.hip);
		// the lane-per-group / workgroup-per-group kernels keep the refinement modes.  ANOFOX_SOLVE_TILES=0: without it.
		static const bool tiles_on = !(getenv("ANOFOX_SOLVE_TILES") && atoi(getenv("ANOFOX_SOLVE_TILES")) == 0);
		// (p <= 10: a 16 x 16 tile is mostly padding and the lane-per-group solve is as fast — 200 000 x 1000 x 9: 0.58 vs 0.70 ms;
		// from there on the tiles win: p = 16 1.74 -> 0.69 ms, 100 000 x 1000 x 32 4.24 -> 0.76 ms, x 24 with inference 3.40 -> 1.77 ms)
		const bool tiles_mid = mid && tiles_on && p >= 11 && solve_tiles_supports((int)p);
		if (hip_fail(tiles_mid ? launch_solve_tiles(a, ss) : solve(0), "wide solve kernel launch", e)) return false;
		// (solve_mid writes complete inference records itself; after the tiles kernel t, p and the interval come from the
		// finish kernel — before the refinement modes, whose final pass rewrites the queued groups' records in full)
		if (tiles_mid && hip_fail(launch_inference_wide_finish(a, ss), "wide inference finish kernel launch", e)) return false;
		// four updates on the wide path: with the residual in double-double every update gains about two digits even at
		// cond(X) = 2e7 (an exactly determined 67 x 67 system of the deep fuzz sweep: 5.4e-7 after two, 4.3e-9 after three,
		// 2.3e-11 after four); the launches are idle unless groups are queued.  ANOFOX_WIDE_REFINE_STEPS overrides (measurements).
		static const int wide_steps = getenv("ANOFOX_WIDE_REFINE_STEPS") ? atoi(getenv("ANOFOX_WIDE_REFINE_STEPS")) : 2 * kRefineSteps;
		for (int it = 0; it < wide_steps; ++it) { // queued groups only: b += (X'WX)^-1 X'Wr
			if (hip_fail(launch_residual_grad_wide(a, ss), "wide residual kernel launch", e)) return false;
			if (hip_fail(solve(1), "wide refine kernel launch", e)) return false;
		}
		if (hip_fail(launch_residual_grad_wide(a, ss), "wide residual kernel launch", e)) return false;
		if (hip_fail(solve(2), "wide final kernel launch", e)) return false;
		if (!mid && hip_fail(launch_inference_wide_finish(a, ss), "wide inference finish kernel launch", e)) return false;
		if (a.hc_type != ANOFOX_HC_NONE && a.inference && a.model != ANOFOX_HIP_MODEL_RIDGE) {
			if (!ensure_buffer(&ctx->aux, &ctx->aux_bytes, (size_t)slab * sizeof(double), "hc scratch", e)) return false;
			a.hc_df = (double *)ctx->aux;
			if (hip_fail(launch_hc_wide(a, ss), "wide hc kernel launch", e)) return false;
		}
		// (r4) LAST: the queued groups once more, from their rows in double-double — standard errors (classical and HC) that do not
		// carry cond(X)^2 eps, and the reference's aliasing rule instead of the 1e-11 pivot test (refit_dd.hip).
		// ANOFOX_REFIT_DD=0: without it (measurements).
		static const bool refit_dd_on = !(getenv("ANOFOX_REFIT_DD") && atoi(getenv("ANOFOX_REFIT_DD")) == 0);
		if (refit_dd_on && hip_fail(launch_refit_dd_wide(a, ss), "double-double refit kernel launch", e)) return false;

		if (overlap && hip_fail(hipEventRecord(ctx->slab_solve_done[buf], ss), "hipEventRecord", e)) return false;
		if (ctx->timing) {
			(void)hipEventRecord(e2, ss);
			ctx->acc_events.emplace_back(e0, e1);
			ctx->solve_events.emplace_back(e1, e2);
		}
	}
	// whatever the caller enqueues next on the main stream comes after the last solves
	if (overlap) {
		for (int b = 0; b < 2 && b < k_slab; ++b)
			if (hip_fail(hipStreamWaitEvent(st, ctx->slab_solve_done[b], 0), "hipStreamWaitEvent", e)) return false;
	}
	return true;
}

// Core of the device path: accumulate -> solve -> (queued groups only) residual RSS -> solve again.
// Everything is enqueued on the context's stream; no host synchronisation.
bool run_device_batch(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const int64_t *d_off,
                      const double *d_y, const double *const *x_cols, const double *d_w,
                      const AnofoxHipBatchOptions &opt, double *d_core, double *d_inf, AnofoxError *e,
                      const int64_t *d_rule_counts = nullptr) {
	if (G == 0) return true;
	if (p > (size_t)kNarrowMaxP)
		return run_wide_batch(ctx, G, p, n_rows, d_off, d_y, x_cols, d_w, opt, d_core, d_inf, d_rule_counts, e);
	Workspace ws;
	if (!carve_workspace(ctx, G, (int)p, &ws, e)) return false;

	BatchArgs a;
	memset(&a, 0, sizeof a);
	a.row_offsets = d_off;
	a.y = d_y;
	for (size_t j = 0; j < p; ++j) a.x[j] = x_cols[j];
	a.w = d_w;
	a.n_groups = G;
	a.n_rows = n_rows;
	a.p = (int)p;
	a.model = (int)opt.model;
	a.fit_intercept = opt.fit_intercept ? 1 : 0;
	a.compute_inference = opt.compute_inference ? 1 : 0;
	a.lambda_scaling = (int)opt.lambda_scaling;
	a.hc_type = (int)opt.hc_type;
	a.confidence_level = opt.confidence_level;
	a.alpha = opt.alpha;
	a.moments = ws.moments;
	a.core = d_core;
	a.inference = opt.compute_inference ? d_inf : nullptr;
	a.refine_list = ws.refine_list;
	a.refine_count = ws.refine_count;
	ctx->last_refine_count = ws.refine_count;
	a.refine_vec = ws.refine_vec;
	a.tcrit_table = ws.tcrit_table;
	a.rule_counts = d_rule_counts;
	a.seg_table = ws.seg_table;
	a.seg_rows = seg_rows_for(n_rows);
	a.row_ends = ctx->frame_ends;

	hipStream_t st = ctx->stream;
	// refine counter + t table + the header of the segment table (contiguous)
	if (hip_fail(hipMemsetAsync(ws.refine_count, 0, 256 + kTcritTableBytes + sizeof(SegHeader), st), "hipMemsetAsync", e)) return false;

	hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
	if (ctx->timing) {
		e0 = get_event(ctx);
		e1 = get_event(ctx);
		e2 = get_event(ctx);
		(void)hipEventRecord(e0, st);
	}
	if (ctx->gate_wait) { // one-shot: the handles belong to the caller and may be gone after this call
		hipEvent_t gw = ctx->gate_wait;
		ctx->gate_wait = nullptr;
		if (hip_fail(hipStreamWaitEvent(st, gw, 0), "hipStreamWaitEvent", e)) return false;
	}
	if (ctx->timing) (void)hipEventRecord(e0, st); // (again: the accumulate kernel starts after the gate)
	// batches of small groups (<= 128 rows on average): several groups per wavefront, the long ones through a list
	static const bool small_on = !(getenv("ANOFOX_ACC_SMALL") && atoi(getenv("ANOFOX_ACC_SMALL")) == 0); // A/B switch
	// (window frames: overlapping row ranges given by row_ends — the packed kernel reads row_offsets[g + 1])
	const int segw = (small_on && !a.row_ends && n_rows > 0 && G < (int64_t)0x7fffffff) ? accumulate_small_segment_width((double)n_rows / (double)G) : 0;
	if (segw) {
		int32_t *big_count = ws.refine_count + 4; // zeroed with the other counters; the list borrows the (still unused) refine queue
		if (hip_fail(launch_accumulate_small(a, segw, ws.refine_list, big_count, st), "accumulate kernel launch", e)) return false;
		if (hip_fail(launch_accumulate_narrow_list(a, ws.refine_list, big_count, st), "accumulate kernel launch", e)) return false;
	} else if (hip_fail(launch_accumulate_narrow(a, st), "accumulate kernel launch", e)) {
		return false;
	}
	if (ctx->timing) (void)hipEventRecord(e1, st);
	if (ctx->gate_record) {
		hipEvent_t gr = ctx->gate_record;
		ctx->gate_record = nullptr;
		if (hip_fail(hipEventRecord(gr, st), "hipEventRecord", e)) return false;
	}
	if (hip_fail(launch_solve_narrow(a, st), "solve kernel launch", e)) return false;
	// queued groups only: kRefineSteps x (b += (X'WX)^-1 X'Wr), then the statistics from the directly summed RSS
	if (hip_fail(launch_refine_fused_narrow(a, kRefineSteps, st), "refine kernel launch", e)) return false;
	// ols.rs:209-231, wls.rs:230-252: HC errors replace the classical ones; ridge has no such branch
	if (a.hc_type != ANOFOX_HC_NONE && a.inference && a.model != ANOFOX_HIP_MODEL_RIDGE) {
		const size_t b_tab = align_up(sizeof(PredictSegTable), 256); // overflow segments of very large groups, then the prep records
		if (!ensure_buffer(&ctx->aux, &ctx->aux_bytes, b_tab + hc_prep_bytes(G, (int)p), "hc scratch", e)) return false;
		if (hip_fail(hipMemsetAsync(ctx->aux, 0, 64, st), "hipMemsetAsync", e)) return false;
		if (hip_fail(launch_hc_narrow(a, (double *)((char *)ctx->aux + b_tab), ctx->aux, st), "hc kernel launch", e)) return false;
	}
	// (r4) LAST: the queued groups once more, from their rows in double-double (refit_dd.hip) — the reference's aliasing rule
	// instead of the 1e-11 pivot test (round 3's narrow sweeps: 12 "rank band" groups of 240 000 cases), standard errors without
	// cond(X)^2 eps.  ANOFOX_REFIT_DD=0: without it.
	static const bool refit_dd_on = !(getenv("ANOFOX_REFIT_DD") && atoi(getenv("ANOFOX_REFIT_DD")) == 0);
	if (refit_dd_on && hip_fail(launch_refit_dd_narrow(a, st), "double-double refit kernel launch", e)) return false;
	if (ctx->timing) {
		(void)hipEventRecord(e2, st);
		ctx->acc_events.emplace_back(e0, e1);   // owns e0 and e1
		ctx->solve_events.emplace_back(e1, e2); // shares e1, owns e2
	}
	return true;
}

bool validate_batch(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const void *off, const void *y,
                    const double *const *x_cols, const void *w, const AnofoxHipBatchOptions &opt, const void *core,
                    const void *inf, AnofoxError *e) {
	if (!ctx) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (G < 0 || n_rows < 0) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "negative n_groups or n_rows"); return false; }
	if (p == 0 || !x_cols) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x is NULL or empty"); return false; }
	if (p > (size_t)kWideMaxP) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT,
		          "n_features = " + std::to_string(p) + " exceeds the supported maximum of " + std::to_string(kWideMaxP));
		return false;
	}
	if (G > 0 && (!off || !y || !core)) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "row_offsets, y or core is NULL"); return false; }
	for (size_t j = 0; j < p; ++j)
		if (G > 0 && !x_cols[j]) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x column pointer is NULL"); return false; }
	if (opt.model != ANOFOX_HIP_MODEL_OLS && opt.model != ANOFOX_HIP_MODEL_RIDGE && opt.model != ANOFOX_HIP_MODEL_WLS) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "unknown model");
		return false;
	}
	if (opt.model == ANOFOX_HIP_MODEL_WLS && G > 0 && !w) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "weights is NULL"); return false; }
	if (opt.compute_inference && G > 0 && !inf) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "inference buffer is NULL"); return false; }
	if ((int)opt.hc_type < ANOFOX_HC_NONE || (int)opt.hc_type > ANOFOX_HC_HC3) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "unknown hc_type");
		return false;
	}
	return true;
}

thread_local std::unique_ptr<AnofoxHipContext, void (*)(AnofoxHipContext *)> tls_ctx(nullptr, anofox_hip_context_destroy);

AnofoxHipContext *default_context(AnofoxError *e) {
	if (!tls_ctx) {
		AnofoxHipContext *c = nullptr;
		if (!anofox_hip_context_create(-1, &c, e)) return nullptr;
		tls_ctx.reset(c);
	}
	return tls_ctx.get();
}

} // namespace

namespace anofox {
namespace host {
bool refit_groups_device(AnofoxHipContext *ctx, int64_t n_groups, size_t p, int64_t n_rows, const int64_t *d_row_offsets,
                         const double *d_y, const double *const *x_cols, const double *d_w, const AnofoxHipBatchOptions &opt,
                         double *d_core, double *d_inf, AnofoxError *e) {
	return run_device_batch(ctx, n_groups, p, n_rows, d_row_offsets, d_y, x_cols, d_w, opt, d_core, d_inf, e);
}
} // namespace host
} // namespace anofox

extern "C" {

const char *anofox_hip_version(void) { return "anofox_stats_hip 0.1 gfx950"; }

size_t anofox_hip_core_record_len(size_t p) { return p + 6; }
size_t anofox_hip_inference_record_len(size_t p) { return 5 * p + 2; }
size_t anofox_hip_max_features(void) { return (size_t)kWideMaxP; }

bool anofox_hip_context_create(int device_id, AnofoxHipContext **out_ctx, AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_ctx is NULL"); return false; }
	*out_ctx = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		(void)hipGetLastError();
		set_error(out_error, ANOFOX_ERROR_INTERNAL, "no HIP device available (this library has no CPU fallback)");
		return false;
	}
	int dev = device_id;
	if (dev < 0) {
		if (hip_fail(hipGetDevice(&dev), "hipGetDevice", out_error)) return false;
	}
	if (dev >= count) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "device_id out of range"); return false; }
	if (hip_fail(hipSetDevice(dev), "hipSetDevice", out_error)) return false;
	auto *ctx = new (std::nothrow) AnofoxHipContext();
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "context allocation failed"); return false; }
	ctx->device = dev;
	if (hip_fail(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking), "hipStreamCreate", out_error)) {
		delete ctx;
		return false;
	}
	ctx->stream = ctx->own_stream;
	*out_ctx = ctx;
	return true;
}

void anofox_hip_context_destroy(AnofoxHipContext *ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	{
		std::vector<AnofoxHipAggState *> states;
		{
			std::lock_guard<std::mutex> lk(ctx->mu);
			states.swap(ctx->agg_states);
		}
		for (auto *st : states) agg_state_detach(st);
	}
	for (auto &pr : ctx->acc_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
	for (auto &pr : ctx->solve_events) { (void)hipEventDestroy(pr.second); }
	for (auto ev : ctx->free_events) (void)hipEventDestroy(ev);
	if (ctx->ws) (void)hipFree(ctx->ws);
	if (ctx->frames_buf) (void)hipFree(ctx->frames_buf);
	if (ctx->stage) (void)hipFree(ctx->stage);
	if (ctx->aux) (void)hipFree(ctx->aux);
	if (ctx->wtab) (void)hipFree(ctx->wtab);
	for (auto &pr : ctx->predict_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
	if (ctx->solve_stream) {
		(void)hipStreamSynchronize(ctx->solve_stream);
		(void)hipStreamDestroy(ctx->solve_stream);
	}
	for (int b = 0; b < 2; ++b) {
		if (ctx->slab_acc_done[b]) (void)hipEventDestroy(ctx->slab_acc_done[b]);
		if (ctx->slab_solve_done[b]) (void)hipEventDestroy(ctx->slab_solve_done[b]);
	}
	if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
	delete ctx;
}

bool anofox_hip_context_set_stream(AnofoxHipContext *ctx, void *hip_stream, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	ctx->stream = (hipStream_t)hip_stream; // NULL = HIP's default stream (torch's default stream), used as given
	return true;
}

bool anofox_hip_context_set_accumulate_gate(AnofoxHipContext *ctx, void *wait_event, void *record_event, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	ctx->gate_wait = (hipEvent_t)wait_event;
	ctx->gate_record = (hipEvent_t)record_event;
	return true;
}

bool anofox_hip_context_use_own_stream(AnofoxHipContext *ctx, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	ctx->stream = ctx->own_stream;
	return true;
}

bool anofox_hip_context_synchronize(AnofoxHipContext *ctx, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	return !hip_fail(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize", out_error);
}

bool anofox_hip_context_enable_timing(AnofoxHipContext *ctx, bool enable, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	ctx->timing = enable;
	return true;
}

bool anofox_hip_context_last_refine_count(AnofoxHipContext *ctx, int64_t *out_count, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx || !out_count) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context or out_count is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	*out_count = 0;
	if (!ctx->last_refine_count) return true;
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	if (hip_fail(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize", out_error)) return false;
	int32_t v = 0;
	if (hip_fail(hipMemcpy(&v, ctx->last_refine_count, sizeof v, hipMemcpyDeviceToHost), "hipMemcpy", out_error)) return false;
	*out_count = v;
	return true;
}

bool anofox_hip_context_last_window_refit_count(AnofoxHipContext *ctx, int64_t *out_count, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx || !out_count) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context or out_count is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	*out_count = ctx->last_window_flagged;
	return true;
}

bool anofox_hip_context_collect_timing(AnofoxHipContext *ctx, AnofoxHipKernelTimes *out, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx || !out) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context or out is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	if (hip_fail(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize", out_error)) return false;
	memset(out, 0, sizeof *out);
	for (auto &pr : ctx->acc_events) {
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
			out->accumulate_ms_min = out->accumulate_count == 0 ? ms : fmin(out->accumulate_ms_min, (double)ms);
			out->accumulate_ms_max = fmax(out->accumulate_ms_max, (double)ms);
			out->accumulate_ms += ms;
			out->accumulate_count++;
		}
	}
	for (auto &pr : ctx->solve_events) {
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { out->solve_ms += ms; out->solve_count++; }
	}
	for (auto &pr : ctx->predict_events) {
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { out->predict_ms += ms; out->predict_count++; }
		ctx->free_events.push_back(pr.first);
		ctx->free_events.push_back(pr.second);
	}
	for (auto &pr : ctx->acc_events) { ctx->free_events.push_back(pr.first); ctx->free_events.push_back(pr.second); }
	for (auto &pr : ctx->solve_events) { ctx->free_events.push_back(pr.second); }
	ctx->acc_events.clear();
	ctx->solve_events.clear();
	ctx->predict_events.clear();
	return true;
}

bool anofox_hip_fit_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                 const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                 const double *d_w, AnofoxHipBatchOptions options, double *d_core, double *d_inference,
                                 AnofoxError *out_error) {
	reset_error(out_error);
	if (!validate_batch(ctx, n_groups, n_features, n_rows, d_row_offsets, d_y, x_cols, d_w, options, d_core,
	                    d_inference, out_error))
		return false;
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const bool ok = run_device_batch(ctx, n_groups, n_features, n_rows, d_row_offsets, d_y, x_cols, d_w, options, d_core,
	                                 d_inference, out_error);
	ctx->gate_wait = ctx->gate_record = nullptr; // the gate never outlives the call it was set for
	return ok;
}


namespace {

// per-row predictions from fit records; uses (and zeroes) the t table at the end of the workspace
bool run_predict(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const int64_t *d_off, const double *const *x_cols,
                 const double *d_core, double confidence, double *d_pred, AnofoxError *e) {
	if (G == 0) return true;
	// scratch: t memo | overflow-segment table (its count sits right behind the memo: one memset) | margin[G]
	const size_t b_tab = align_up(sizeof(PredictSegTable), 256);
	if (!ensure_buffer(&ctx->aux, &ctx->aux_bytes, kTcritTableBytes + b_tab + (size_t)G * sizeof(double), "predict scratch", e)) return false;
	PredictArgs a;
	memset(&a, 0, sizeof a);
	a.row_offsets = d_off;
	for (size_t j = 0; j < p; ++j) a.x_table[j] = x_cols[j];
	a.core = d_core;
	a.pred = d_pred;
	a.n_groups = G;
	a.p = (int)p;
	a.confidence_level = confidence;
	a.tcrit_table = ctx->aux;
	a.seg_table = (char *)ctx->aux + kTcritTableBytes;
	a.seg_rows = seg_rows_for(n_rows);
	a.avg_rows = (n_rows > 0 && G > 0) ? (double)n_rows / (double)G : 0.0;
	a.margin = (double *)((char *)ctx->aux + kTcritTableBytes + b_tab);
	if (hip_fail(hipMemsetAsync(ctx->aux, 0, kTcritTableBytes + 64, ctx->stream), "hipMemsetAsync", e)) return false;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (ctx->timing) {
		e0 = get_event(ctx);
		e1 = get_event(ctx);
		(void)hipEventRecord(e0, ctx->stream);
	}
	if (hip_fail(launch_predict(a, ctx->stream), "predict kernel launch", e)) return false;
	if (ctx->timing) {
		(void)hipEventRecord(e1, ctx->stream);
		ctx->predict_events.emplace_back(e0, e1);
	}
	return true;
}

} // namespace

bool anofox_hip_predict_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                     const int64_t *d_row_offsets, const double *const *x_cols, const double *d_core,
                                     double confidence_level, double *d_pred, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (n_groups < 0 || n_features == 0 || n_features > (size_t)kWideMaxP || !x_cols) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "invalid n_groups / n_features / x");
		return false;
	}
	if (n_groups > 0 && (!d_row_offsets || !d_core || !d_pred)) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets, core or pred is NULL");
		return false;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	return run_predict(ctx, n_groups, n_features, n_rows, d_row_offsets, x_cols, d_core, confidence_level, d_pred, out_error);
}

bool anofox_hip_information_criteria_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, const double *d_core,
                                                  AnofoxHipBatchOptions options, double *d_out, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (n_groups < 0 || n_features == 0 || n_features > (size_t)kWideMaxP) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "invalid n_groups / n_features");
		return false;
	}
	if (n_groups > 0 && (!d_core || !d_out)) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "core or out is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	return !hip_fail(launch_information_criteria(d_core, n_groups, (int)n_features, options.fit_intercept ? 1 : 0,
	                                             options.model == ANOFOX_HIP_MODEL_WLS ? 1 : 0, d_out, ctx->stream),
	                 "information criteria kernel launch", out_error);
}

bool anofox_hip_information_criteria_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, const double *core,
                                                AnofoxHipBatchOptions options, double *out, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) {
		ctx = default_context(out_error);
		if (!ctx) return false;
	}
	if (n_groups < 0 || n_features == 0 || n_features > (size_t)kWideMaxP) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "invalid n_groups / n_features");
		return false;
	}
	if (n_groups == 0) return true;
	if (!core || !out) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "core or out is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const size_t G = (size_t)n_groups, b_core = align_up(G * (n_features + 6) * sizeof(double), 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_core + G * 3 * sizeof(double), "staging", out_error)) return false;
	double *d_core = (double *)ctx->stage, *d_out = (double *)((char *)ctx->stage + b_core);
	hipStream_t st = ctx->stream;
	if (hip_fail(hipMemcpyAsync(d_core, core, G * (n_features + 6) * sizeof(double), hipMemcpyHostToDevice, st), "H2D core", out_error)) return false;
	if (hip_fail(launch_information_criteria(d_core, n_groups, (int)n_features, options.fit_intercept ? 1 : 0,
	                                         options.model == ANOFOX_HIP_MODEL_WLS ? 1 : 0, d_out, st),
	             "information criteria kernel launch", out_error))
		return false;
	if (hip_fail(hipMemcpyAsync(out, d_out, G * 3 * sizeof(double), hipMemcpyDeviceToHost, st), "D2H", out_error)) return false;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

bool anofox_hip_fit_predict_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                         const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                         const double *d_w, const int64_t *d_train_counts, AnofoxHipBatchOptions options,
                                         double *d_core, double *d_pred, AnofoxError *out_error) {
	reset_error(out_error);
	options.compute_inference = false; // ols_predict_aggregate.cpp:353
	if (!validate_batch(ctx, n_groups, n_features, n_rows, d_row_offsets, d_y, x_cols, d_w, options, d_core, nullptr,
	                    out_error))
		return false;
	if (n_groups > 0 && !d_pred) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "pred is NULL"); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	if (!run_device_batch(ctx, n_groups, n_features, n_rows, d_row_offsets, d_y, x_cols, d_w, options, d_core, nullptr,
	                      out_error, d_train_counts))
		return false;
	return run_predict(ctx, n_groups, n_features, n_rows, d_row_offsets, x_cols, d_core, options.confidence_level, d_pred,
	                   out_error);
}

bool anofox_hip_fit_predict_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                       const int64_t *row_offsets, const double *y, const double *const *x_cols,
                                       const double *w, const int64_t *train_counts, AnofoxHipBatchOptions options,
                                       double *core, double *pred, AnofoxError *out_error) {
	reset_error(out_error);
	options.compute_inference = false;
	if (!ctx) {
		ctx = default_context(out_error);
		if (!ctx) return false;
	}
	if (!validate_batch(ctx, n_groups, n_features, n_rows, row_offsets, y, x_cols, w, options, core, nullptr, out_error))
		return false;
	if (n_groups == 0) return true;
	if (!pred) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "pred is NULL"); return false; }
	if (row_offsets[0] != 0 || row_offsets[n_groups] != n_rows) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must start at 0 and end at n_rows");
		return false;
	}
	for (int64_t g = 0; g < n_groups; ++g)
		if (row_offsets[g + 1] < row_offsets[g]) {
			set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must be non-decreasing");
			return false;
		}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const size_t p = n_features;
	const bool weighted = options.model == ANOFOX_HIP_MODEL_WLS;
	const size_t ncol = p + 1 + (weighted ? 1 : 0);
	const size_t core_len = p + 6;
	const size_t R = (size_t)n_rows, G = (size_t)n_groups;
	const size_t b_off = align_up((G + 1) * sizeof(int64_t), 256);
	const size_t b_cnt = train_counts ? align_up(G * sizeof(int64_t), 256) : 0;
	const size_t b_col = align_up((R + 2) * sizeof(double), 256);
	const size_t b_core = align_up(G * core_len * sizeof(double), 256);
	const size_t b_pred = align_up(R * 3 * sizeof(double) + 8, 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_off + b_cnt + ncol * b_col + b_core + b_pred, "staging", out_error))
		return false;
	char *cur = (char *)ctx->stage;
	hipStream_t st = ctx->stream;
	int64_t *d_off = (int64_t *)cur;
	cur += b_off;
	if (hip_fail(hipMemcpyAsync(d_off, row_offsets, (G + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D offsets", out_error)) return false;
	int64_t *d_cnt = nullptr;
	if (train_counts) {
		d_cnt = (int64_t *)cur;
		cur += b_cnt;
		if (hip_fail(hipMemcpyAsync(d_cnt, train_counts, G * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D counts", out_error)) return false;
	}
	const double *d_x[kWideMaxP];
	for (size_t j = 0; j < p; ++j) {
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, x_cols[j], R * sizeof(double), hipMemcpyHostToDevice, st), "H2D x", out_error)) return false;
		d_x[j] = (const double *)cur;
		cur += b_col;
	}
	if (R > 0 && hip_fail(hipMemcpyAsync(cur, y, R * sizeof(double), hipMemcpyHostToDevice, st), "H2D y", out_error)) return false;
	const double *d_y = (const double *)cur;
	cur += b_col;
	const double *d_w = nullptr;
	if (weighted) {
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, w, R * sizeof(double), hipMemcpyHostToDevice, st), "H2D w", out_error)) return false;
		d_w = (const double *)cur;
		cur += b_col;
	}
	double *d_core = (double *)cur;
	cur += b_core;
	double *d_pred = (double *)cur;
	if (!run_device_batch(ctx, n_groups, p, n_rows, d_off, d_y, d_x, d_w, options, d_core, nullptr, out_error, d_cnt)) return false;
	if (!run_predict(ctx, n_groups, p, n_rows, d_off, d_x, d_core, options.confidence_level, d_pred, out_error)) return false;
	if (hip_fail(hipMemcpyAsync(core, d_core, G * core_len * sizeof(double), hipMemcpyDeviceToHost, st), "D2H core", out_error)) return false;
	if (R > 0 && hip_fail(hipMemcpyAsync(pred, d_pred, R * 3 * sizeof(double), hipMemcpyDeviceToHost, st), "D2H pred", out_error)) return false;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}


namespace {

// ---- window frames as virtual groups (frames.hip): any width, any frame, with the fit path's refinement ----
constexpr int64_t kFrameSlab = 1 << 20; // frames fitted per pass (bounds the record scratch: (p + 6) doubles each)

struct FrameScratch {
	int64_t *ynn;   // [n_rows + 1]
	void *scan_tmp;
	size_t scan_tmp_bytes;
	int64_t *lo, *hi; // [n_rows] (frames computed from a ROWS spec), else nullptr
	int64_t *rule;  // [slab]
	double *core;   // [slab * (p + 6)]
};

bool carve_frames(AnofoxHipContext *ctx, int64_t n_rows, int64_t n_frames, size_t p, bool need_bounds, FrameScratch *fs, AnofoxError *e) {
	const int64_t slab = n_frames < kFrameSlab ? n_frames : kFrameSlab;
	const size_t b_ynn = align_up(((size_t)n_rows + 2) * sizeof(int64_t), 256);
	const size_t b_tmp = align_up(frames_scan_temp_bytes(n_rows) + 256, 256);
	const size_t b_bnd = need_bounds ? align_up((size_t)n_rows * sizeof(int64_t), 256) : 0;
	const size_t b_rule = align_up((size_t)slab * sizeof(int64_t), 256);
	const size_t b_core = align_up((size_t)slab * (p + 6) * sizeof(double), 256);
	if (!ensure_buffer(&ctx->frames_buf, &ctx->frames_bytes, b_ynn + b_tmp + 2 * b_bnd + b_rule + b_core, "window frame scratch", e)) return false;
	char *base = (char *)ctx->frames_buf;
	fs->ynn = (int64_t *)base;
	fs->scan_tmp = base + b_ynn;
	fs->scan_tmp_bytes = b_tmp;
	fs->lo = need_bounds ? (int64_t *)(base + b_ynn + b_tmp) : nullptr;
	fs->hi = need_bounds ? (int64_t *)(base + b_ynn + b_tmp + b_bnd) : nullptr;
	fs->rule = (int64_t *)(base + b_ynn + b_tmp + 2 * b_bnd);
	fs->core = (double *)(base + b_ynn + b_tmp + 2 * b_bnd + b_rule);
	return true;
}

// Fit every frame [d_lo[e], d_hi[e]) as a group and predict its last row into d_pred[(d_list ? d_list[e] : e)].
bool run_frames(AnofoxHipContext *ctx, const FrameScratch &fs, int64_t n_frames, size_t p, int64_t n_rows, const double *d_y,
                const double *const *x_cols, const double *d_w, const int64_t *d_lo, const int64_t *d_hi, AnofoxHipBatchOptions opt,
                double *d_pred, const int32_t *d_list, AnofoxError *e) {
	if (n_frames == 0) return true;
	opt.compute_inference = false; // the window functions fit without inference (ols_fit_predict.cpp:296)
	opt.hc_type = ANOFOX_HC_NONE;
	if (!ensure_buffer(&ctx->wtab, &ctx->wtab_bytes, (size_t)(kWindowTcritCap + 1) * sizeof(double), "t table", e)) return false;
	hipStream_t st = ctx->stream;
	if (ctx->wtab_conf != opt.confidence_level) {
		if (hip_fail(launch_tcrit_table((double *)ctx->wtab, kWindowTcritCap, 0.5 * (1.0 + opt.confidence_level), st), "t table kernel launch", e)) return false;
		ctx->wtab_conf = opt.confidence_level;
	}
	FrameArgs a;
	memset(&a, 0, sizeof a);
	a.y = d_y;
	for (size_t j = 0; j < p; ++j) a.x_table[j] = x_cols[j];
	a.p = (int)p;
	a.fit_intercept = opt.fit_intercept ? 1 : 0;
	a.confidence_level = opt.confidence_level;
	a.ynn = fs.ynn;
	a.rule_counts = fs.rule;
	a.core = fs.core;
	a.tcrit = (const double *)ctx->wtab;
	a.tcrit_cap = kWindowTcritCap;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (ctx->timing) {
		e0 = get_event(ctx);
		e1 = get_event(ctx);
		(void)hipEventRecord(e0, st);
	}
	const bool was_timing = ctx->timing;
	ctx->timing = false; // the passes below are reported as one "predict" interval, not as accumulate / solve launches
	bool ok = true;
	for (int64_t s0 = 0; ok && s0 < n_frames; s0 += kFrameSlab) {
		const int64_t S = n_frames - s0 < kFrameSlab ? n_frames - s0 : kFrameSlab;
		a.n_frames = S;
		a.lo = d_lo + s0;
		a.hi = d_hi + s0;
		a.pred = d_list ? d_pred : d_pred + 3 * s0;
		a.list = d_list ? d_list + s0 : nullptr;
		ok = !hip_fail(launch_frames_rule(a, st), "frame rule kernel launch", e);
		ctx->frame_ends = a.hi;
		ok = ok && run_device_batch(ctx, S, p, n_rows, a.lo, d_y, x_cols, d_w, opt, fs.core, nullptr, e, fs.rule);
		ctx->frame_ends = nullptr;
		ok = ok && !hip_fail(launch_frames_predict(a, st), "frame predict kernel launch", e);
	}
	ctx->timing = was_timing;
	if (ctx->timing) {
		(void)hipEventRecord(e1, st);
		ctx->predict_events.emplace_back(e0, e1);
	}
	return ok;
}

bool run_window(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const int64_t *d_off, const double *d_y,
                const double *const *x_cols, const double *d_w, const AnofoxHipWindowFrame &frame,
                const AnofoxHipBatchOptions &opt, double *d_pred, AnofoxError *e) {
	if (G == 0) return true;
	if (p > (size_t)kNarrowMaxP) {
		// wider than the in-register window kernels: every frame becomes a virtual group of the batch fit (frames.hip)
		if (n_rows == 0) return true;
		FrameScratch fs;
		if (!carve_frames(ctx, n_rows, n_rows, p, true, &fs, e)) return false;
		hipStream_t st = ctx->stream;
		if (hip_fail(launch_frames_ynn(d_y, n_rows, fs.ynn, fs.scan_tmp, fs.scan_tmp_bytes, st), "frame scan launch", e)) return false;
		if (hip_fail(launch_frames_from_rows_spec(d_off, G, n_rows, frame.start_preceding, frame.end_preceding, fs.lo, fs.hi, st),
		             "frame bounds kernel launch", e))
			return false;
		ctx->frame_prefix = frame.start_preceding == ANOFOX_HIP_FRAME_UNBOUNDED; // (every frame starts at its partition's first row)
		const bool ok = run_frames(ctx, fs, n_rows, p, n_rows, d_y, x_cols, d_w, fs.lo, fs.hi, opt, d_pred, nullptr, e);
		ctx->frame_prefix = false;
		return ok;
	}
	if (!ensure_buffer(&ctx->wtab, &ctx->wtab_bytes, (size_t)(kWindowTcritCap + 1) * sizeof(double), "t table", e)) return false;
	hipStream_t st = ctx->stream;
	if (ctx->wtab_conf != opt.confidence_level) {
		if (hip_fail(launch_tcrit_table((double *)ctx->wtab, kWindowTcritCap, 0.5 * (1.0 + opt.confidence_level), st), "t table kernel launch", e)) return false;
		ctx->wtab_conf = opt.confidence_level;
	}
	WindowArgs a;
	memset(&a, 0, sizeof a);
	a.row_offsets = d_off;
	a.y = d_y;
	for (size_t j = 0; j < p; ++j) a.x[j] = x_cols[j];
	a.w = d_w;
	a.pred = d_pred;
	a.n_groups = G;
	a.p = (int)p;
	a.model = (int)opt.model;
	a.fit_intercept = opt.fit_intercept ? 1 : 0;
	a.lambda_scaling = (int)opt.lambda_scaling;
	a.alpha = opt.alpha;
	a.tcrit = (const double *)ctx->wtab;
	a.tcrit_cap = kWindowTcritCap;
	a.frame_start = frame.start_preceding;
	a.frame_end = frame.end_preceding;
	a.avg_rows = (n_rows > 0 && G > 0) ? (double)n_rows / (double)G : 0.0;
	// ill-conditioned frames are flagged by the kernel and refitted below with the fit path's refinement passes
	static const bool flag_on = !(getenv("ANOFOX_WIN_FLAG") && atoi(getenv("ANOFOX_WIN_FLAG")) == 0); // A/B switch for measurements
	const bool flagging = flag_on && n_rows > 0 && n_rows < (int64_t)0x7fffffff;
	if (flagging) {
		if (!ensure_buffer(&ctx->aux, &ctx->aux_bytes, 256 + (size_t)n_rows * sizeof(int32_t), "window flag list", e)) return false;
		a.flag_count = (int32_t *)ctx->aux;
		a.flag_list = (int32_t *)((char *)ctx->aux + 256);
		a.flag_cap = (int32_t)n_rows;
		if (hip_fail(hipMemsetAsync(a.flag_count, 0, sizeof(int32_t), st), "hipMemsetAsync", e)) return false;
	}
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (ctx->timing) {
		e0 = get_event(ctx);
		e1 = get_event(ctx);
		(void)hipEventRecord(e0, st);
	}
	if (hip_fail(launch_window_predict(a, st), "window kernel launch", e)) return false;
	if (ctx->timing) {
		(void)hipEventRecord(e1, st);
		ctx->predict_events.emplace_back(e0, e1);
	}
	if (!flagging) return true;
	// (the one host synchronisation of this path: how many frames were flagged — usually none)
	int32_t n_flag = 0;
	if (hip_fail(hipMemcpyAsync(&n_flag, a.flag_count, sizeof n_flag, hipMemcpyDeviceToHost, st), "D2H flag count", e)) return false;
	if (hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", e)) return false;
	ctx->last_window_flagged = n_flag;
	if (n_flag <= 0) return true;
	if (n_flag > a.flag_cap) n_flag = a.flag_cap;
	FrameScratch fs;
	if (!carve_frames(ctx, n_rows, n_flag, p, true, &fs, e)) return false;
	if ((int64_t)n_flag * 64 < n_rows) fs.ynn = nullptr; // few frames: their training rows are counted directly
	else if (hip_fail(launch_frames_ynn(d_y, n_rows, fs.ynn, fs.scan_tmp, fs.scan_tmp_bytes, st), "frame scan launch", e)) return false;
	if (hip_fail(launch_frames_from_rows_spec(d_off, G, n_rows, frame.start_preceding, frame.end_preceding, fs.lo, fs.hi, st, a.flag_list, n_flag),
	             "frame bounds kernel launch", e))
		return false;
	return run_frames(ctx, fs, n_flag, p, n_rows, d_y, x_cols, d_w, fs.lo, fs.hi, opt, d_pred, a.flag_list, e);
}

bool validate_window(AnofoxHipContext *ctx, int64_t G, size_t p, const void *off, const void *y, const double *const *x_cols,
                     const void *w, const AnofoxHipWindowFrame &frame, const AnofoxHipBatchOptions &opt, const void *pred,
                     AnofoxError *e) {
	if (!ctx) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (frame.start_preceding < frame.end_preceding || frame.start_preceding == -ANOFOX_HIP_FRAME_UNBOUNDED ||
	    frame.end_preceding == ANOFOX_HIP_FRAME_UNBOUNDED) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "window frame must start at or before its end");
		return false;
	}
	if (G < 0 || p == 0 || !x_cols) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "invalid n_groups / n_features / x"); return false; }
	if (p > (size_t)kWideMaxP) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "the window path supports at most " + std::to_string(kWideMaxP) + " features");
		return false;
	}
	if (G > 0 && (!off || !y || !pred)) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "row_offsets, y or pred is NULL"); return false; }
	for (size_t j = 0; j < p; ++j)
		if (G > 0 && !x_cols[j]) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x column pointer is NULL"); return false; }
	if (opt.model == ANOFOX_HIP_MODEL_WLS && G > 0 && !w) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "weights is NULL"); return false; }
	return true;
}

} // namespace

bool anofox_hip_fit_predict_window_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                          const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                          const double *d_w, AnofoxHipWindowFrame frame, AnofoxHipBatchOptions options,
                                          double *d_pred, AnofoxError *out_error) {
	reset_error(out_error);
	(void)n_rows;
	if (!validate_window(ctx, n_groups, n_features, d_row_offsets, d_y, x_cols, d_w, frame, options, d_pred, out_error)) return false;
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	return run_window(ctx, n_groups, n_features, n_rows, d_row_offsets, d_y, x_cols, d_w, frame, options, d_pred, out_error);
}

namespace {
bool validate_frames(AnofoxHipContext *ctx, int64_t n_rows, size_t p, const void *y, const double *const *x_cols, const void *w,
                     const void *lo, const void *hi, const AnofoxHipBatchOptions &opt, const void *pred, AnofoxError *e) {
	if (!ctx) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (n_rows < 0 || p == 0 || p > (size_t)kWideMaxP || !x_cols) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "invalid n_rows / n_features / x"); return false; }
	if (n_rows > 0 && (!y || !lo || !hi || !pred)) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "y, frame bounds or pred is NULL"); return false; }
	for (size_t j = 0; j < p; ++j)
		if (n_rows > 0 && !x_cols[j]) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x column pointer is NULL"); return false; }
	if (opt.model != ANOFOX_HIP_MODEL_OLS && opt.model != ANOFOX_HIP_MODEL_RIDGE && opt.model != ANOFOX_HIP_MODEL_WLS) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT, "unknown model");
		return false;
	}
	if (opt.model == ANOFOX_HIP_MODEL_WLS && n_rows > 0 && !w) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "weights is NULL"); return false; }
	return true;
}
} // namespace

bool anofox_hip_fit_predict_frames_device(AnofoxHipContext *ctx, int64_t n_rows, size_t n_features, const double *d_y,
                                          const double *const *x_cols, const double *d_w, const int64_t *d_frame_lo,
                                          const int64_t *d_frame_hi, AnofoxHipBatchOptions options, double *d_pred,
                                          AnofoxError *out_error) {
	reset_error(out_error);
	if (!validate_frames(ctx, n_rows, n_features, d_y, x_cols, d_w, d_frame_lo, d_frame_hi, options, d_pred, out_error)) return false;
	if (n_rows == 0) return true;
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	FrameScratch fs;
	if (!carve_frames(ctx, n_rows, n_rows, n_features, false, &fs, out_error)) return false;
	if (hip_fail(launch_frames_ynn(d_y, n_rows, fs.ynn, fs.scan_tmp, fs.scan_tmp_bytes, ctx->stream), "frame scan launch", out_error)) return false;
	return run_frames(ctx, fs, n_rows, n_features, n_rows, d_y, x_cols, d_w, d_frame_lo, d_frame_hi, options, d_pred, nullptr, out_error);
}

bool anofox_hip_fit_predict_frames_host(AnofoxHipContext *ctx, int64_t n_rows, size_t n_features, const double *y,
                                        const double *const *x_cols, const double *w, const int64_t *frame_lo,
                                        const int64_t *frame_hi, AnofoxHipBatchOptions options, double *pred, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) {
		ctx = default_context(out_error);
		if (!ctx) return false;
	}
	if (!validate_frames(ctx, n_rows, n_features, y, x_cols, w, frame_lo, frame_hi, options, pred, out_error)) return false;
	if (n_rows == 0) return true;
	for (int64_t e = 0; e < n_rows; ++e)
		if (frame_lo[e] < 0 || frame_hi[e] > n_rows || (frame_hi[e] > frame_lo[e] && frame_lo[e] >= n_rows)) {
			set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "frame bounds must lie within [0, n_rows]");
			return false;
		}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const size_t p = n_features, R = (size_t)n_rows;
	const bool weighted = options.model == ANOFOX_HIP_MODEL_WLS;
	const size_t ncol = p + 1 + (weighted ? 1 : 0);
	const size_t b_col = align_up((R + 2) * sizeof(double), 256), b_bnd = align_up(R * sizeof(int64_t), 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, ncol * b_col + 2 * b_bnd + align_up(R * 3 * sizeof(double) + 8, 256), "staging", out_error))
		return false;
	char *cur = (char *)ctx->stage;
	hipStream_t st = ctx->stream;
	const double *d_x[kWideMaxP];
	for (size_t j = 0; j < p; ++j) {
		if (hip_fail(hipMemcpyAsync(cur, x_cols[j], R * sizeof(double), hipMemcpyHostToDevice, st), "H2D x", out_error)) return false;
		d_x[j] = (const double *)cur;
		cur += b_col;
	}
	if (hip_fail(hipMemcpyAsync(cur, y, R * sizeof(double), hipMemcpyHostToDevice, st), "H2D y", out_error)) return false;
	const double *d_y = (const double *)cur;
	cur += b_col;
	const double *d_w = nullptr;
	if (weighted) {
		if (hip_fail(hipMemcpyAsync(cur, w, R * sizeof(double), hipMemcpyHostToDevice, st), "H2D w", out_error)) return false;
		d_w = (const double *)cur;
		cur += b_col;
	}
	int64_t *d_lo = (int64_t *)cur;
	cur += b_bnd;
	int64_t *d_hi = (int64_t *)cur;
	cur += b_bnd;
	double *d_pred = (double *)cur;
	if (hip_fail(hipMemcpyAsync(d_lo, frame_lo, R * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D frame_lo", out_error)) return false;
	if (hip_fail(hipMemcpyAsync(d_hi, frame_hi, R * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D frame_hi", out_error)) return false;
	FrameScratch fs;
	if (!carve_frames(ctx, n_rows, n_rows, p, false, &fs, out_error)) return false;
	if (hip_fail(launch_frames_ynn(d_y, n_rows, fs.ynn, fs.scan_tmp, fs.scan_tmp_bytes, st), "frame scan launch", out_error)) return false;
	if (!run_frames(ctx, fs, n_rows, p, n_rows, d_y, d_x, d_w, d_lo, d_hi, options, d_pred, nullptr, out_error)) return false;
	if (hip_fail(hipMemcpyAsync(pred, d_pred, R * 3 * sizeof(double), hipMemcpyDeviceToHost, st), "D2H pred", out_error)) return false;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

bool anofox_hip_fit_predict_expanding_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                             const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                             const double *d_w, AnofoxHipBatchOptions options, double *d_pred,
                                             AnofoxError *out_error) {
	const AnofoxHipWindowFrame frame = {ANOFOX_HIP_FRAME_UNBOUNDED, 0}; // UNBOUNDED PRECEDING .. CURRENT ROW
	return anofox_hip_fit_predict_window_device(ctx, n_groups, n_features, n_rows, d_row_offsets, d_y, x_cols, d_w, frame,
	                                            options, d_pred, out_error);
}

bool anofox_hip_fit_predict_expanding_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                           const int64_t *row_offsets, const double *y, const double *const *x_cols,
                                           const double *w, AnofoxHipBatchOptions options, double *pred,
                                           AnofoxError *out_error) {
	const AnofoxHipWindowFrame frame = {ANOFOX_HIP_FRAME_UNBOUNDED, 0}; // UNBOUNDED PRECEDING .. CURRENT ROW
	return anofox_hip_fit_predict_window_host(ctx, n_groups, n_features, n_rows, row_offsets, y, x_cols, w, frame, options, pred,
	                                          out_error);
}

bool anofox_hip_fit_predict_window_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                        const int64_t *row_offsets, const double *y, const double *const *x_cols,
                                        const double *w, AnofoxHipWindowFrame frame, AnofoxHipBatchOptions options,
                                        double *pred, AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) {
		ctx = default_context(out_error);
		if (!ctx) return false;
	}
	if (!validate_window(ctx, n_groups, n_features, row_offsets, y, x_cols, w, frame, options, pred, out_error)) return false;
	if (n_groups == 0) return true;
	if (row_offsets[0] != 0 || row_offsets[n_groups] != n_rows) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must start at 0 and end at n_rows");
		return false;
	}
	for (int64_t g = 0; g < n_groups; ++g)
		if (row_offsets[g + 1] < row_offsets[g]) {
			set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must be non-decreasing");
			return false;
		}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const size_t p = n_features, R = (size_t)n_rows, G = (size_t)n_groups;
	const bool weighted = options.model == ANOFOX_HIP_MODEL_WLS;
	const size_t ncol = p + 1 + (weighted ? 1 : 0);
	const size_t b_off = align_up((G + 1) * sizeof(int64_t), 256);
	const size_t b_col = align_up((R + 2) * sizeof(double), 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_off + ncol * b_col + align_up(R * 3 * sizeof(double) + 8, 256), "staging", out_error))
		return false;
	char *cur = (char *)ctx->stage;
	hipStream_t st = ctx->stream;
	int64_t *d_off = (int64_t *)cur;
	cur += b_off;
	if (hip_fail(hipMemcpyAsync(d_off, row_offsets, (G + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D offsets", out_error)) return false;
	const double *d_x[kWideMaxP];
	for (size_t j = 0; j < p; ++j) {
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, x_cols[j], R * sizeof(double), hipMemcpyHostToDevice, st), "H2D x", out_error)) return false;
		d_x[j] = (const double *)cur;
		cur += b_col;
	}
	if (R > 0 && hip_fail(hipMemcpyAsync(cur, y, R * sizeof(double), hipMemcpyHostToDevice, st), "H2D y", out_error)) return false;
	const double *d_y = (const double *)cur;
	cur += b_col;
	const double *d_w = nullptr;
	if (weighted) {
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, w, R * sizeof(double), hipMemcpyHostToDevice, st), "H2D w", out_error)) return false;
		d_w = (const double *)cur;
		cur += b_col;
	}
	double *d_pred = (double *)cur;
	if (!run_window(ctx, n_groups, p, n_rows, d_off, d_y, d_x, d_w, frame, options, d_pred, out_error)) return false;
	if (R > 0 && hip_fail(hipMemcpyAsync(pred, d_pred, R * 3 * sizeof(double), hipMemcpyDeviceToHost, st), "D2H pred", out_error)) return false;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

bool anofox_hip_fit_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                               const int64_t *row_offsets, const double *y, const double *const *x_cols,
                               const double *w, AnofoxHipBatchOptions options, double *core, double *inference,
                               AnofoxError *out_error) {
	reset_error(out_error);
	if (!ctx) {
		ctx = default_context(out_error);
		if (!ctx) return false;
	}
	if (!validate_batch(ctx, n_groups, n_features, n_rows, row_offsets, y, x_cols, w, options, core, inference,
	                    out_error))
		return false;
	if (n_groups == 0) return true;
	for (int64_t g = 0; g < n_groups; ++g) {
		if (row_offsets[g + 1] < row_offsets[g] || row_offsets[g] < 0 || row_offsets[g + 1] > n_rows) {
			set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must be non-decreasing and within [0, n_rows]");
			return false;
		}
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;

	const size_t p = n_features;
	const bool weighted = options.model == ANOFOX_HIP_MODEL_WLS;
	const size_t ncol = p + 1 + (weighted ? 1 : 0);
	const size_t core_len = p + 6, inf_len = 5 * p + 2;
	// stream the groups through the GPU in slabs of at most ~32M rows
	const int64_t slab_rows = 32ll << 20;
	std::vector<int64_t> off;
	int64_t g0 = 0;
	while (g0 < n_groups) {
		int64_t g1 = g0 + 1;
		while (g1 < n_groups && row_offsets[g1 + 1] - row_offsets[g0] <= slab_rows) ++g1;
		const int64_t G = g1 - g0;
		const int64_t r0 = row_offsets[g0], r1 = row_offsets[g1];
		const int64_t R = r1 - r0;
		off.resize((size_t)G + 1);
		for (int64_t g = 0; g <= G; ++g) off[(size_t)g] = row_offsets[g0 + g] - r0;

		const size_t b_off = align_up(((size_t)G + 1) * sizeof(int64_t), 256);
		const size_t b_col = align_up(((size_t)R + 2) * sizeof(double), 256);
		const size_t b_core = align_up((size_t)G * core_len * sizeof(double), 256);
		const size_t b_inf = options.compute_inference ? align_up((size_t)G * inf_len * sizeof(double), 256) : 0;
		const size_t total = b_off + ncol * b_col + b_core + b_inf;
		if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, total, "staging", out_error)) return false;
		char *base = (char *)ctx->stage;
		int64_t *d_off = (int64_t *)base;
		char *cur = base + b_off;
		hipStream_t st = ctx->stream;
		if (hip_fail(hipMemcpyAsync(d_off, off.data(), ((size_t)G + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D offsets", out_error)) return false;
		const double *d_x[kWideMaxP];
		for (size_t j = 0; j < p; ++j) {
			if (R > 0 && hip_fail(hipMemcpyAsync(cur, x_cols[j] + r0, (size_t)R * sizeof(double), hipMemcpyHostToDevice, st), "H2D x", out_error)) return false;
			d_x[j] = (const double *)cur;
			cur += b_col;
		}
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, y + r0, (size_t)R * sizeof(double), hipMemcpyHostToDevice, st), "H2D y", out_error)) return false;
		const double *d_y = (const double *)cur;
		cur += b_col;
		const double *d_w = nullptr;
		if (weighted) {
			if (R > 0 && hip_fail(hipMemcpyAsync(cur, w + r0, (size_t)R * sizeof(double), hipMemcpyHostToDevice, st), "H2D w", out_error)) return false;
			d_w = (const double *)cur;
			cur += b_col;
		}
		double *d_core = (double *)cur;
		cur += b_core;
		double *d_inf = options.compute_inference ? (double *)cur : nullptr;
		if (!run_device_batch(ctx, G, p, R, d_off, d_y, d_x, d_w, options, d_core, d_inf, out_error)) return false;
		if (hip_fail(hipMemcpyAsync(core + (size_t)g0 * core_len, d_core, (size_t)G * core_len * sizeof(double), hipMemcpyDeviceToHost, st), "D2H core", out_error)) return false;
		if (d_inf && hip_fail(hipMemcpyAsync(inference + (size_t)g0 * inf_len, d_inf, (size_t)G * inf_len * sizeof(double), hipMemcpyDeviceToHost, st), "D2H inference", out_error)) return false;
		if (hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error)) return false;
		g0 = g1;
	}
	return true;
}

} // extern "C"

/* ------------------------------------------------------------------------------------------------ */
/* Reference-compatible single-group symbols                                                          */
/* ------------------------------------------------------------------------------------------------ */

namespace {

// DataArray::to_vec (types.rs:79-89): NULL -> NaN
void expand(const AnofoxDataArray &a, std::vector<double> &out) {
	out.resize(a.len);
	for (size_t i = 0; i < a.len; ++i) {
		const bool valid = !a.validity || ((a.validity[i / 8] >> (i % 8)) & 1);
		out[i] = valid ? a.data[i] : NAN;
	}
}

std::string fmt_g(double v) {
	char buf[64];
	snprintf(buf, sizeof buf, "%g", v);
	return buf;
}

void default_inference(AnofoxFitResultInference *inf) { // FitResultInference::default(), types.rs:151-165
	memset(inf, 0, sizeof *inf);
	inf->confidence_level = 0.95;
	inf->f_statistic = NAN;
	inf->f_pvalue = NAN;
}

bool fit_single(const char *what, AnofoxDataArray y, const AnofoxDataArray *x, size_t x_count, const AnofoxDataArray *weights,
                AnofoxHipBatchOptions opt, AnofoxFitResultCore *out_core, AnofoxFitResultInference *out_inference,
                AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_core) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_core is NULL"); return false; }          // lib.rs:113-118
	if (!x || x_count == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "x is NULL or empty"); return false; } // lib.rs:120-125
	// core-crate validation order: ridge alpha first (ridge.rs:38-40), then emptiness and lengths (ols.rs:38-56, wls.rs:44-73)
	if (opt.model == ANOFOX_HIP_MODEL_RIDGE && opt.alpha < 0.0) {
		set_error(out_error, ANOFOX_ERROR_INVALID_ALPHA, "Invalid alpha parameter: " + fmt_g(opt.alpha) + " (must be >= 0)");
		return false;
	}
	if (y.len == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Empty input: y cannot be empty"); return false; }
	if (weights) {
		if (weights->len == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Empty input: weights cannot be empty"); return false; }
		if (weights->len != y.len) {
			set_error(out_error, ANOFOX_ERROR_DIMENSION_MISMATCH, "Dimension mismatch: y has " + std::to_string(y.len) + " elements, X has " + std::to_string(weights->len) + " rows");
			return false;
		}
	}
	for (size_t j = 0; j < x_count; ++j) {
		if (x[j].len != y.len) {
			set_error(out_error, ANOFOX_ERROR_DIMENSION_MISMATCH, "Dimension mismatch: y has " + std::to_string(y.len) + " elements, X has " + std::to_string(x[j].len) + " rows");
			return false;
		}
	}
	const size_t p = x_count, n = y.len;
	if (p > (size_t)kWideMaxP) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, std::string(what) + ": more than " + std::to_string(kWideMaxP) + " features are not supported by the GPU path");
		return false;
	}
	// The kernels apply the aggregate's "< 2 rows -> NULL" rule from row_offsets; a single-group call has no
	// such rule (ols.rs accepts n == 1), so a one-row input is padded with one all-NaN row, which the row
	// filter drops again.
	const size_t n_pad = n < 2 ? 2 : n;
	std::vector<std::vector<double>> cols(p);
	std::vector<double> yv, wv;
	expand(y, yv);
	yv.resize(n_pad, NAN);
	for (size_t j = 0; j < p; ++j) { expand(x[j], cols[j]); cols[j].resize(n_pad, NAN); }
	if (weights) { expand(*weights, wv); wv.resize(n_pad, NAN); }

	std::vector<const double *> xp(p);
	for (size_t j = 0; j < p; ++j) xp[j] = cols[j].data();
	const int64_t off[2] = {0, (int64_t)n_pad};
	std::vector<double> core(p + 6), inf(5 * p + 2);
	if (!anofox_hip_fit_batch_host(nullptr, 1, p, (int64_t)n_pad, off, yv.data(), xp.data(), weights ? wv.data() : nullptr, opt,
	                               core.data(), inf.data(), out_error))
		return false;
	const int status = (int)core[p + 5];
	if (status != ANOFOX_ERROR_SUCCESS) {
		size_t n_valid = 0;
		for (size_t i = 0; i < n; ++i) {
			bool ok = isfinite(yv[i]);
			for (size_t j = 0; ok && j < p; ++j) ok = isfinite(cols[j][i]);
			if (ok && weights) ok = wv[i] > 0.0 && isfinite(wv[i]);
			n_valid += ok;
		}
		std::string msg;
		switch (status) { // error strings: crates/anofox-stats-core/src/errors.rs:5-59
		case ANOFOX_ERROR_NO_VALID_DATA: msg = "All rows filtered due to NULL/NaN values"; break;
		case ANOFOX_ERROR_INSUFFICIENT_DATA:
			msg = "Insufficient data: " + std::to_string(n_valid) + " rows, " + std::to_string(p) + " features (need rows > features)";
			break;
		case ANOFOX_ERROR_INVALID_ALPHA: msg = "Invalid alpha parameter: " + fmt_g(opt.alpha) + " (must be >= 0)"; break;
		default: msg = std::string(what) + " failed on the GPU path"; break;
		}
		set_error(out_error, (AnofoxErrorCode)status, msg);
		return false;
	}
	double *coef = (double *)malloc(p * sizeof(double));
	if (!coef) { set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "Failed to allocate coefficients"); return false; }
	memcpy(coef, core.data(), p * sizeof(double));
	double *arr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	// the reference returns inference: None for the intercept-only shortcut even when requested (ols.rs:128);
	// that shortcut is the only successful fit whose coefficients are all NaN
	bool any_coef = false;
	for (size_t j = 0; j < p; ++j) any_coef = any_coef || !isnan(core[j]);
	const bool got_inf = out_inference != nullptr && opt.compute_inference && any_coef;
	if (got_inf) {
		for (int k = 0; k < 5; ++k) {
			arr[k] = (double *)malloc(p * sizeof(double));
			if (!arr[k]) {
				for (int m = 0; m < k; ++m) free(arr[m]);
				free(coef);
				set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "Failed to allocate inference arrays");
				return false;
			}
			memcpy(arr[k], inf.data() + (size_t)k * p, p * sizeof(double));
		}
	}
	out_core->coefficients = coef;
	out_core->coefficients_len = p;
	out_core->intercept = core[p];
	out_core->r_squared = core[p + 1];
	out_core->adj_r_squared = core[p + 2];
	out_core->residual_std_error = core[p + 3];
	out_core->n_observations = (size_t)core[p + 4];
	out_core->n_features = p;
	if (out_inference) {
		if (got_inf) {
			out_inference->std_errors = arr[0];
			out_inference->t_values = arr[1];
			out_inference->p_values = arr[2];
			out_inference->ci_lower = arr[3];
			out_inference->ci_upper = arr[4];
			out_inference->len = p;
			out_inference->confidence_level = opt.confidence_level;
			out_inference->f_statistic = inf[5 * p];
			out_inference->f_pvalue = inf[5 * p + 1];
		} else {
			default_inference(out_inference);
		}
	}
	return true;
}

} // namespace

extern "C" {

bool anofox_ols_fit(AnofoxDataArray y, const AnofoxDataArray *x, size_t x_count, AnofoxOlsOptions options,
                    AnofoxFitResultCore *out_core, AnofoxFitResultInference *out_inference, AnofoxError *out_error) {
	AnofoxHipBatchOptions o;
	memset(&o, 0, sizeof o);
	o.model = ANOFOX_HIP_MODEL_OLS;
	o.fit_intercept = options.fit_intercept;
	o.compute_inference = options.compute_inference;
	o.confidence_level = options.confidence_level;
	o.solver = options.solver;
	o.hc_type = options.hc_type;
	return fit_single("OLS fit", y, x, x_count, nullptr, o, out_core, out_inference, out_error);
}

bool anofox_ridge_fit(AnofoxDataArray y, const AnofoxDataArray *x, size_t x_count, AnofoxRidgeOptions options,
                      AnofoxFitResultCore *out_core, AnofoxFitResultInference *out_inference, AnofoxError *out_error) {
	AnofoxHipBatchOptions o;
	memset(&o, 0, sizeof o);
	o.model = ANOFOX_HIP_MODEL_RIDGE;
	o.alpha = options.alpha;
	o.fit_intercept = options.fit_intercept;
	o.compute_inference = options.compute_inference;
	o.confidence_level = options.confidence_level;
	o.solver = options.solver;
	o.lambda_scaling = options.lambda_scaling;
	o.hc_type = ANOFOX_HC_NONE;
	return fit_single("Ridge fit", y, x, x_count, nullptr, o, out_core, out_inference, out_error);
}

bool anofox_wls_fit(AnofoxDataArray y, const AnofoxDataArray *x, size_t x_count, AnofoxDataArray weights,
                    AnofoxWlsOptions options, AnofoxFitResultCore *out_core, AnofoxFitResultInference *out_inference,
                    AnofoxError *out_error) {
	AnofoxHipBatchOptions o;
	memset(&o, 0, sizeof o);
	o.model = ANOFOX_HIP_MODEL_WLS;
	o.fit_intercept = options.fit_intercept;
	o.compute_inference = options.compute_inference;
	o.confidence_level = options.confidence_level;
	o.solver = options.solver;
	o.hc_type = options.hc_type;
	return fit_single("WLS fit", y, x, x_count, &weights, o, out_core, out_inference, out_error);
}

void anofox_free_result_core(AnofoxFitResultCore *result) {
	if (!result) return;
	if (result->coefficients) {
		free(result->coefficients);
		result->coefficients = nullptr;
	}
}

void anofox_free_result_inference(AnofoxFitResultInference *result) {
	if (!result) return;
	double **arr[5] = {&result->std_errors, &result->t_values, &result->p_values, &result->ci_lower, &result->ci_upper};
	for (auto pp : arr) {
		if (*pp) {
			free(*pp);
			*pp = nullptr;
		}
	}
}

bool anofox_compute_aic(double rss, size_t n, size_t k, double *out_aic, AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_aic) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_aic is NULL"); return false; }
	if (n == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Invalid input: n must be > 0"); return false; }
	if (rss < 0.0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Invalid input: RSS must be non-negative"); return false; }
	*out_aic = rss == 0.0 ? -INFINITY : (double)n * log(rss / (double)n) + 2.0 * (double)k;
	return true;
}

bool anofox_compute_bic(double rss, size_t n, size_t k, double *out_bic, AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_bic) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_bic is NULL"); return false; }
	if (n == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Invalid input: n must be > 0"); return false; }
	if (rss < 0.0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Invalid input: RSS must be non-negative"); return false; }
	*out_bic = rss == 0.0 ? -INFINITY : (double)n * log(rss / (double)n) + (double)k * log((double)n);
	return true;
}

double anofox_t_critical(double confidence_level, size_t df) {
	if (df == 0 || !(confidence_level > 0.0) || !(confidence_level < 1.0)) return NAN; // lib.rs:2218-2220
	return hostmath::t_quantile_upper(0.5 * (1.0 + confidence_level), (double)df);
}

bool anofox_predict_with_interval(const double *coefficients, size_t coefficients_len, double intercept,
                                  const double *x_new, size_t x_len, double residual_std_error, size_t n_observations,
                                  double confidence_level, AnofoxPredictionResult *out_result) {
	if (!out_result) return false;
	out_result->yhat = out_result->yhat_lower = out_result->yhat_upper = NAN;
	if (!coefficients || coefficients_len == 0) return false;
	if (!x_new || x_len != coefficients_len) return false;
	double yhat = isnan(intercept) ? 0.0 : intercept;
	for (size_t j = 0; j < coefficients_len; ++j)
		if (!isnan(coefficients[j])) yhat += coefficients[j] * x_new[j]; // NaN coefficients contribute 0
	out_result->yhat = out_result->yhat_lower = out_result->yhat_upper = yhat;
	if (isnan(residual_std_error) || residual_std_error <= 0.0 || n_observations <= coefficients_len + 1) return true;
	const bool has_icpt = !isnan(intercept);
	const size_t used = coefficients_len + (has_icpt ? 1 : 0);
	const size_t df = n_observations > used ? n_observations - used : 0;
	if (df == 0) return true;
	const double tcrit = anofox_t_critical(confidence_level, df);
	if (isnan(tcrit)) return true;
	const double n = (double)n_observations;
	const double margin = tcrit * residual_std_error * sqrt(1.0 + 1.0 / n);
	out_result->yhat_lower = yhat - margin;
	out_result->yhat_upper = yhat + margin;
	return true;
}

bool anofox_predict(const AnofoxDataArray *x, size_t x_count, const double *coefficients, size_t coefficients_len,
                    double intercept, double **out_predictions, size_t *out_predictions_len, AnofoxError *out_error) {
	reset_error(out_error);
	if (!x || x_count == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "x is NULL or empty"); return false; }
	if (!coefficients || coefficients_len == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "coefficients is NULL or empty"); return false; }
	if (!out_predictions || !out_predictions_len) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_predictions or out_predictions_len is NULL");
		return false;
	}
	if (x_count != coefficients_len) {
		set_error(out_error, ANOFOX_ERROR_DIMENSION_MISMATCH, "Dimension mismatch: y has " + std::to_string(coefficients_len) + " elements, X has " + std::to_string(x_count) + " rows");
		return false;
	}
	const size_t n = x[0].len, p = x_count;
	for (size_t j = 0; j < p; ++j)
		if (x[j].len != n) {
			set_error(out_error, ANOFOX_ERROR_DIMENSION_MISMATCH, "Dimension mismatch: y has " + std::to_string(n) + " elements, X has " + std::to_string(x[j].len) + " rows");
			return false;
		}
	if (p > (size_t)kWideMaxP) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "too many features for the GPU path"); return false; }
	double *out = (double *)malloc((n ? n : 1) * sizeof(double));
	if (!out) { set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "Failed to allocate predictions"); return false; }
	if (n > 0) {
		AnofoxHipContext *ctx = default_context(out_error);
		if (!ctx) { free(out); return false; }
		std::lock_guard<std::mutex> lk(ctx->mu);
		bool ok = !hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error);
		// one group holding every row; a fit record carrying the coefficients (n_obs = 0 -> no interval)
		std::vector<std::vector<double>> cols(p);
		std::vector<double> core(p + 6, 0.0), pred(3 * n);
		bool nan_coef = false;
		for (size_t j = 0; j < p; ++j) { expand(x[j], cols[j]); core[j] = coefficients[j]; nan_coef = nan_coef || isnan(coefficients[j]); }
		core[p] = intercept;
		core[p + 3] = NAN;
		const size_t b_col = align_up(n * sizeof(double), 256);
		const size_t total = 256 + p * b_col + align_up(core.size() * sizeof(double), 256) + 3 * n * sizeof(double);
		ok = ok && ensure_buffer(&ctx->stage, &ctx->stage_bytes, total, "staging", out_error);
		if (ok) {
			char *cur = (char *)ctx->stage;
			hipStream_t st = ctx->stream;
			const int64_t off[2] = {0, (int64_t)n};
			int64_t *d_off = (int64_t *)cur;
			cur += 256;
			ok = !hip_fail(hipMemcpyAsync(d_off, off, sizeof off, hipMemcpyHostToDevice, st), "H2D", out_error);
			const double *d_x[kWideMaxP];
			for (size_t j = 0; ok && j < p; ++j) {
				ok = !hip_fail(hipMemcpyAsync(cur, cols[j].data(), n * sizeof(double), hipMemcpyHostToDevice, st), "H2D", out_error);
				d_x[j] = (const double *)cur;
				cur += b_col;
			}
			double *d_core = (double *)cur;
			cur += align_up(core.size() * sizeof(double), 256);
			double *d_pred = (double *)cur;
			ok = ok && !hip_fail(hipMemcpyAsync(d_core, core.data(), core.size() * sizeof(double), hipMemcpyHostToDevice, st), "H2D", out_error);
			ok = ok && run_predict(ctx, 1, p, (int64_t)n, d_off, d_x, d_core, 0.95, d_pred, out_error);
			ok = ok && !hip_fail(hipMemcpyAsync(pred.data(), d_pred, 3 * n * sizeof(double), hipMemcpyDeviceToHost, st), "D2H", out_error);
			ok = ok && !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
		}
		if (!ok) { free(out); return false; }
		// upstream's plain predict lets a NaN coefficient poison every prediction (predict.rs:55-61)
		for (size_t i = 0; i < n; ++i) out[i] = nan_coef ? NAN : pred[3 * i];
	}
	*out_predictions = out;
	*out_predictions_len = n;
	return true;
}

void anofox_free_predictions(double *predictions) {
	if (predictions) free(predictions);
}

} // extern "C"

/* ------------------------------------------------------------------------------------------------ */
/* Variance inflation factors (vif_agg / vif): p OLS fits per group on one Gram matrix                */
/* ------------------------------------------------------------------------------------------------ */

namespace {

// d_vif[g] = { vif[p], status }.  min_rows = 3 for the aggregate's rule (vif_aggregate.cpp:154), 0 for compute_vif.
bool run_vif(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const int64_t *d_off, const double *const *x_cols,
             int64_t min_rows, double *d_vif, AnofoxError *e) {
	if (G == 0) return true;
	hipStream_t st = ctx->stream;
	if (p <= (size_t)kNarrowMaxP) {
		Workspace ws;
		if (!carve_workspace(ctx, G, (int)p, &ws, e)) return false;
		BatchArgs a;
		memset(&a, 0, sizeof a);
		a.row_offsets = d_off;
		a.y = x_cols[0]; // the y slot of the record is not used
		for (size_t j = 0; j < p; ++j) a.x[j] = x_cols[j];
		a.n_groups = G;
		a.n_rows = n_rows;
		a.p = (int)p;
		a.model = ANOFOX_HIP_MODEL_OLS;
		a.fit_intercept = 1;
		a.moments = ws.moments;
		a.seg_table = ws.seg_table;
		a.seg_rows = seg_rows_for(n_rows);
		if (hip_fail(hipMemsetAsync(ws.seg_table, 0, sizeof(SegHeader), st), "hipMemsetAsync", e)) return false;
		hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
		if (ctx->timing) {
			e0 = get_event(ctx);
			e1 = get_event(ctx);
			e2 = get_event(ctx);
			(void)hipEventRecord(e0, st);
		}
		if (hip_fail(launch_accumulate_narrow(a, st), "accumulate kernel launch", e)) return false;
		if (ctx->timing) (void)hipEventRecord(e1, st);
		if (hip_fail(launch_vif_narrow(ws.moments, d_off, G, (int)p, min_rows, d_vif, st), "vif kernel launch", e)) return false;
		if (ctx->timing) {
			(void)hipEventRecord(e2, st);
			ctx->acc_events.emplace_back(e0, e1);
			ctx->solve_events.emplace_back(e1, e2);
		}
		return true;
	}
	// wide designs: one grouped fit per feature (x_j on the others); only R^2 and the status of each fit are used
	const size_t q = p - 1;
	if (!ensure_buffer(&ctx->aux, &ctx->aux_bytes, (size_t)G * (q + 6) * sizeof(double), "vif scratch", e)) return false;
	double *d_core = (double *)ctx->aux;
	AnofoxHipBatchOptions opt;
	memset(&opt, 0, sizeof opt);
	opt.model = ANOFOX_HIP_MODEL_OLS;
	opt.fit_intercept = true;
	opt.confidence_level = 0.95;
	std::vector<const double *> others(q);
	for (size_t j = 0; j < p; ++j) {
		for (size_t i = 0, k = 0; i < p; ++i)
			if (i != j) others[k++] = x_cols[i];
		if (!run_device_batch(ctx, G, q, n_rows, d_off, x_cols[j], others.data(), nullptr, opt, d_core, nullptr, e)) return false;
		if (hip_fail(launch_vif_from_core(d_core, d_off, G, (int)q, (int)j, (int)p, min_rows, d_vif, st), "vif kernel launch", e)) return false;
	}
	return true;
}

bool validate_vif(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const void *off, const double *const *x_cols,
                  const void *out, AnofoxError *e) {
	if (!ctx) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (G < 0 || n_rows < 0) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "negative n_groups or n_rows"); return false; }
	if (p == 0 || !x_cols) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x is NULL or empty"); return false; }
	if (p > (size_t)kWideMaxP + 1) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT,
		          "n_features = " + std::to_string(p) + " exceeds the supported maximum of " + std::to_string(kWideMaxP + 1));
		return false;
	}
	if (G > 0 && (!off || !out)) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "row_offsets or output is NULL"); return false; }
	for (size_t j = 0; j < p; ++j)
		if (G > 0 && !x_cols[j]) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x column pointer is NULL"); return false; }
	return true;
}

bool vif_host(AnofoxHipContext *ctx, int64_t n_groups, size_t p, int64_t n_rows, const int64_t *row_offsets,
              const double *const *x_cols, int64_t min_rows, double *vif, AnofoxError *out_error) {
	if (!ctx) {
		ctx = default_context(out_error);
		if (!ctx) return false;
	}
	if (!validate_vif(ctx, n_groups, p, n_rows, row_offsets, x_cols, vif, out_error)) return false;
	if (n_groups == 0) return true;
	if (row_offsets[0] != 0 || row_offsets[n_groups] != n_rows) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must start at 0 and end at n_rows");
		return false;
	}
	for (int64_t g = 0; g < n_groups; ++g)
		if (row_offsets[g + 1] < row_offsets[g]) {
			set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must be non-decreasing");
			return false;
		}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const size_t R = (size_t)n_rows, G = (size_t)n_groups;
	const size_t b_off = align_up((G + 1) * sizeof(int64_t), 256);
	const size_t b_col = align_up((R + 2) * sizeof(double), 256);
	const size_t b_out = align_up(G * (p + 1) * sizeof(double), 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_off + p * b_col + b_out, "staging", out_error)) return false;
	char *cur = (char *)ctx->stage;
	hipStream_t st = ctx->stream;
	int64_t *d_off = (int64_t *)cur;
	cur += b_off;
	if (hip_fail(hipMemcpyAsync(d_off, row_offsets, (G + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D offsets", out_error)) return false;
	std::vector<const double *> d_x(p);
	for (size_t j = 0; j < p; ++j) {
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, x_cols[j], R * sizeof(double), hipMemcpyHostToDevice, st), "H2D x", out_error)) return false;
		d_x[j] = (const double *)cur;
		cur += b_col;
	}
	double *d_vif = (double *)cur;
	if (!run_vif(ctx, n_groups, p, n_rows, d_off, d_x.data(), min_rows, d_vif, out_error)) return false;
	if (hip_fail(hipMemcpyAsync(vif, d_vif, G * (p + 1) * sizeof(double), hipMemcpyDeviceToHost, st), "D2H vif", out_error)) return false;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

} // namespace

extern "C" {

size_t anofox_hip_vif_record_len(size_t n_features) { return n_features + 1; }
size_t anofox_hip_vif_max_features(void) { return (size_t)kWideMaxP + 1; }

bool anofox_hip_vif_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                 const int64_t *d_row_offsets, const double *const *x_cols, double *d_vif,
                                 AnofoxError *out_error) {
	reset_error(out_error);
	if (!validate_vif(ctx, n_groups, n_features, n_rows, d_row_offsets, x_cols, d_vif, out_error)) return false;
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	return run_vif(ctx, n_groups, n_features, n_rows, d_row_offsets, x_cols, 3, d_vif, out_error);
}

bool anofox_hip_vif_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                               const int64_t *row_offsets, const double *const *x_cols, double *vif, AnofoxError *out_error) {
	reset_error(out_error);
	return vif_host(ctx, n_groups, n_features, n_rows, row_offsets, x_cols, 3, vif, out_error);
}

// lib.rs:1688-1739 over vif.rs:23-98
bool anofox_compute_vif(const AnofoxDataArray *x, size_t x_count, double **out_vif, size_t *out_vif_len, AnofoxError *out_error) {
	reset_error(out_error);
	if (!x || x_count == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "x is NULL or empty"); return false; }
	if (!out_vif || !out_vif_len) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_vif or out_vif_len is NULL"); return false; }
	double *res = (double *)malloc(x_count * sizeof(double));
	if (!res) { set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "Failed to allocate VIF array"); return false; }
	if (x_count == 1) { // vif.rs:30-33
		res[0] = 1.0;
		*out_vif = res;
		*out_vif_len = 1;
		return true;
	}
	const size_t n = x[0].len;
	if (n == 0) { free(res); set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Invalid input: Empty feature arrays"); return false; }
	for (size_t j = 0; j < x_count; ++j)
		if (x[j].len != n) {
			free(res);
			set_error(out_error, ANOFOX_ERROR_DIMENSION_MISMATCH, "Dimension mismatch: Feature " + std::to_string(j) + " has " +
			          std::to_string(x[j].len) + " observations, expected " + std::to_string(n));
			return false;
		}
	std::vector<std::vector<double>> cols(x_count);
	std::vector<const double *> ptrs(x_count);
	for (size_t j = 0; j < x_count; ++j) {
		expand(x[j], cols[j]);
		ptrs[j] = cols[j].data();
	}
	const int64_t off[2] = {0, (int64_t)n};
	std::vector<double> rec(x_count + 1);
	if (!vif_host(nullptr, 1, x_count, (int64_t)n, off, ptrs.data(), 0, rec.data(), out_error)) { free(res); return false; }
	memcpy(res, rec.data(), x_count * sizeof(double));
	*out_vif = res;
	*out_vif_len = x_count;
	return true;
}

void anofox_free_vif(double *vif) {
	if (vif) free(vif);
}

} // extern "C"

/* ===================================================================================================== */
/* Residual diagnostics (residuals_diagnostics_agg / residuals_diagnostics / anofox_compute_residuals)   */
/* ===================================================================================================== */
namespace {

bool validate_residuals(AnofoxHipContext *ctx, int64_t G, size_t p, int64_t n_rows, const void *off, const void *y,
                        const void *y_hat, const double *const *x_cols, const void *out, const void *group, AnofoxError *e) {
	if (!ctx) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "context is NULL"); return false; }
	if (G < 0 || n_rows < 0) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "negative n_groups or n_rows"); return false; }
	if (p > (size_t)kResidualsMaxP) {
		set_error(e, ANOFOX_ERROR_INVALID_INPUT,
		          "n_features = " + std::to_string(p) + " exceeds the supported maximum of " + std::to_string(kResidualsMaxP) +
		              " for residual diagnostics");
		return false;
	}
	if (p > 0 && !x_cols) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x is NULL"); return false; }
	if (G > 0 && (!off || !group)) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "row_offsets or group output is NULL"); return false; }
	if (n_rows > 0 && (!y || !y_hat || !out)) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "y, y_hat or output is NULL"); return false; }
	for (size_t j = 0; j < p; ++j)
		if (n_rows > 0 && !x_cols[j]) { set_error(e, ANOFOX_ERROR_INVALID_INPUT, "x column pointer is NULL"); return false; }
	return true;
}

bool run_residuals(AnofoxHipContext *ctx, int64_t G, size_t p, const int64_t *d_off, const double *d_y, const double *d_y_hat,
                   const double *const *x_cols, const double *d_rse, bool include_studentized, bool drop_nan_rows,
                   double *d_out, double *d_group, AnofoxError *e) {
	if (G == 0) return true;
	ResidualArgs a{};
	a.row_offsets = d_off;
	a.y = d_y;
	a.y_hat = d_y_hat;
	for (size_t j = 0; j < p && j < (size_t)kNarrowMaxP; ++j) a.x[j] = x_cols[j];
	a.rse = d_rse;
	a.out = d_out;
	a.group_out = d_group;
	a.n_groups = G;
	a.p = (int)p;
	a.include_studentized = include_studentized ? 1 : 0;
	a.drop_nan_rows = drop_nan_rows ? 1 : 0;
	if (p > (size_t)kNarrowMaxP) {
		// 9 .. 128 features: workgroup-per-group kernels; their column pointers travel through a small device table
		if (!ensure_buffer(&ctx->aux, &ctx->aux_bytes, (size_t)kResidualsMaxP * sizeof(double *), "residual column table", e)) return false;
		// (x_cols is pageable host memory: the runtime has read it into its staging buffer when the call returns)
		if (hip_fail(hipMemcpyAsync(ctx->aux, x_cols, p * sizeof(double *), hipMemcpyHostToDevice, ctx->stream), "H2D column table", e)) return false;
		return !hip_fail(launch_residuals_wide(a, (const double *const *)ctx->aux, ctx->stream), "residuals kernel launch", e);
	}
	return !hip_fail(launch_residuals_narrow(a, ctx->stream), "residuals kernel launch", e);
}

bool residuals_host(AnofoxHipContext *ctx, int64_t n_groups, size_t p, int64_t n_rows, const int64_t *row_offsets,
                    const double *y, const double *y_hat, const double *const *x_cols, const double *rse,
                    bool include_studentized, bool drop_nan_rows, double *out, double *group, AnofoxError *out_error) {
	if (!ctx) {
		ctx = default_context(out_error);
		if (!ctx) return false;
	}
	if (!validate_residuals(ctx, n_groups, p, n_rows, row_offsets, y, y_hat, x_cols, out, group, out_error)) return false;
	if (n_groups == 0) return true;
	if (row_offsets[0] != 0 || row_offsets[n_groups] != n_rows) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must start at 0 and end at n_rows");
		return false;
	}
	for (int64_t g = 0; g < n_groups; ++g)
		if (row_offsets[g + 1] < row_offsets[g]) {
			set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "row_offsets must be non-decreasing");
			return false;
		}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	const size_t R = (size_t)n_rows, G = (size_t)n_groups;
	const size_t b_off = align_up((G + 1) * sizeof(int64_t), 256);
	const size_t b_col = align_up((R + 2) * sizeof(double), 256);
	const size_t b_grp = align_up(G * 2 * sizeof(double), 256);
	const size_t b_out = align_up(R * 4 * sizeof(double) + 8, 256);
	if (!ensure_buffer(&ctx->stage, &ctx->stage_bytes, b_off + (p + 2) * b_col + 2 * b_grp + b_out, "staging", out_error)) return false;
	char *cur = (char *)ctx->stage;
	hipStream_t st = ctx->stream;
	int64_t *d_off = (int64_t *)cur;
	cur += b_off;
	if (hip_fail(hipMemcpyAsync(d_off, row_offsets, (G + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st), "H2D offsets", out_error)) return false;
	const double *src[2] = {y, y_hat};
	const double *d_yy[2];
	for (int k = 0; k < 2; ++k) {
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, src[k], R * sizeof(double), hipMemcpyHostToDevice, st), "H2D y", out_error)) return false;
		d_yy[k] = (const double *)cur;
		cur += b_col;
	}
	std::vector<const double *> d_x(p);
	for (size_t j = 0; j < p; ++j) {
		if (R > 0 && hip_fail(hipMemcpyAsync(cur, x_cols[j], R * sizeof(double), hipMemcpyHostToDevice, st), "H2D x", out_error)) return false;
		d_x[j] = (const double *)cur;
		cur += b_col;
	}
	double *d_rse = nullptr;
	if (rse) {
		d_rse = (double *)cur;
		if (hip_fail(hipMemcpyAsync(d_rse, rse, G * sizeof(double), hipMemcpyHostToDevice, st), "H2D rse", out_error)) return false;
	}
	cur += b_grp;
	double *d_group = (double *)cur;
	cur += b_grp;
	double *d_out = (double *)cur;
	if (!run_residuals(ctx, n_groups, p, d_off, d_yy[0], d_yy[1], d_x.data(), d_rse, include_studentized, drop_nan_rows, d_out,
	                   d_group, out_error))
		return false;
	if (R > 0 && hip_fail(hipMemcpyAsync(out, d_out, R * 4 * sizeof(double), hipMemcpyDeviceToHost, st), "D2H residuals", out_error)) return false;
	if (hip_fail(hipMemcpyAsync(group, d_group, G * 2 * sizeof(double), hipMemcpyDeviceToHost, st), "D2H residual groups", out_error)) return false;
	return !hip_fail(hipStreamSynchronize(st), "hipStreamSynchronize", out_error);
}

} // namespace

extern "C" {

size_t anofox_hip_residuals_max_features(void) { return (size_t)kResidualsMaxP; }

bool anofox_hip_residuals_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                       const int64_t *d_row_offsets, const double *d_y, const double *d_y_hat,
                                       const double *const *x_cols, const double *d_rse, bool include_studentized,
                                       bool drop_nan_rows, double *d_out, double *d_group, AnofoxError *out_error) {
	reset_error(out_error);
	if (!validate_residuals(ctx, n_groups, n_features, n_rows, d_row_offsets, d_y, d_y_hat, x_cols, d_out, d_group, out_error))
		return false;
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	return run_residuals(ctx, n_groups, n_features, d_row_offsets, d_y, d_y_hat, x_cols, d_rse, include_studentized,
	                     drop_nan_rows, d_out, d_group, out_error);
}

bool anofox_hip_residuals_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                     const int64_t *row_offsets, const double *y, const double *y_hat,
                                     const double *const *x_cols, const double *rse, bool include_studentized,
                                     bool drop_nan_rows, double *out, double *group, AnofoxError *out_error) {
	reset_error(out_error);
	return residuals_host(ctx, n_groups, n_features, n_rows, row_offsets, y, y_hat, x_cols, rse, include_studentized,
	                      drop_nan_rows, out, group, out_error);
}

// lib.rs:1787-1894 over residuals.rs:30-145
bool anofox_compute_residuals(AnofoxDataArray y, AnofoxDataArray y_hat, const AnofoxDataArray *x, size_t x_count,
                              double residual_std_error, bool include_studentized, AnofoxResidualsResult *out_result,
                              AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_result) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_result is NULL"); return false; }
	*out_result = AnofoxResidualsResult{};
	const size_t n = y.len;
	if (n == 0) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "Invalid input: Empty y array"); return false; } // residuals.rs:39-41
	if (y_hat.len != n) { // residuals.rs:43-49
		set_error(out_error, ANOFOX_ERROR_DIMENSION_MISMATCH,
		          "Dimension mismatch: y has " + std::to_string(n) + " elements, y_hat has " + std::to_string(y_hat.len));
		return false;
	}
	const size_t p = (x && x_count > 0) ? x_count : 0;
	std::vector<double> yv, yh;
	expand(y, yv);
	expand(y_hat, yh);
	std::vector<std::vector<double>> cols(p);
	std::vector<const double *> ptrs(p);
	for (size_t j = 0; j < p; ++j) {
		if (include_studentized && x[j].len != n) { // the reference indexes x[j][i] for i < n and panics on a short column
			set_error(out_error, ANOFOX_ERROR_DIMENSION_MISMATCH, "Dimension mismatch: Feature " + std::to_string(j) + " has " +
			          std::to_string(x[j].len) + " observations, expected " + std::to_string(n));
			return false;
		}
		expand(x[j], cols[j]);
		cols[j].resize(n, NAN);
		ptrs[j] = cols[j].data();
	}
	const int64_t off[2] = {0, (int64_t)n};
	std::vector<double> rec(n * 4);
	double grp[2] = {0.0, 0.0};
	const double rse = residual_std_error;
	if (!residuals_host(nullptr, 1, p, (int64_t)n, off, yv.data(), yh.data(), ptrs.data(), &rse, include_studentized, false,
	                    rec.data(), grp, out_error))
		return false;
	const int flags = (int)grp[1];
	const bool has[4] = {true, (flags & ANOFOX_HIP_RESIDUALS_HAS_STANDARDIZED) != 0, (flags & ANOFOX_HIP_RESIDUALS_HAS_STUDENTIZED) != 0,
	                     (flags & ANOFOX_HIP_RESIDUALS_HAS_LEVERAGE) != 0};
	double *arr[4] = {nullptr, nullptr, nullptr, nullptr};
	for (int k = 0; k < 4; ++k) {
		if (!has[k]) continue;
		arr[k] = (double *)malloc(n * sizeof(double));
		if (!arr[k]) {
			for (int m = 0; m < k; ++m) free(arr[m]);
			set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "Failed to allocate residuals");
			return false;
		}
		for (size_t i = 0; i < n; ++i) arr[k][i] = rec[i * 4 + k];
	}
	out_result->raw = arr[0];
	out_result->standardized = arr[1];
	out_result->studentized = arr[2];
	out_result->leverage = arr[3];
	out_result->len = n;
	out_result->has_standardized = has[1];
	out_result->has_studentized = has[2];
	out_result->has_leverage = has[3];
	return true;
}

void anofox_free_residuals(AnofoxResidualsResult *result) {
	if (!result) return;
	free(result->raw);
	free(result->standardized);
	free(result->studentized);
	free(result->leverage);
	result->raw = result->standardized = result->studentized = result->leverage = nullptr;
	result->len = 0;
	result->has_standardized = result->has_studentized = result->has_leverage = false;
}

} // extern "C"
