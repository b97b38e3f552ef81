// context.h — the context object and the small host-side helpers shared by the translation units that implement
// the C ABI (host_api.hip, agg_state.hip).  Internal: nothing here is exported.
#pragma once
#include <math.h>
#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

struct AnofoxHipContext {
	int device = 0;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	std::mutex mu;
	// device workspace (moments, refine queue, direct RSS)
	void *ws = nullptr;
	size_t ws_bytes = 0;
	// device staging for the host-pointer entry points
	void *stage = nullptr;
	size_t stage_bytes = 0;
	// small auxiliary device buffer (t-quantile memo of the predict kernel)
	void *aux = nullptr;
	size_t aux_bytes = 0;
	// run_wide_batch with several slabs of groups: the solve / refinement kernels of slab k run on this stream while the
	// main stream accumulates slab k + 1 (two moment buffers); events per buffer parity
	hipStream_t solve_stream = nullptr;
	hipEvent_t slab_acc_done[2] = {nullptr, nullptr}, slab_solve_done[2] = {nullptr, nullptr};
	const int32_t *last_refine_count = nullptr; // device address of the most recent launch's queue counter
	hipEvent_t gate_wait = nullptr, gate_record = nullptr; // anofox_hip_context_set_accumulate_gate
	// set around a fit call by the window-frame path (frames.hip): group g owns rows [row_offsets[g], frame_ends[g])
	const int64_t *frame_ends = nullptr;
	bool frame_prefix = false; // (r4) the frames of this call extend one another (UNBOUNDED PRECEDING): accumulate_prefix.hip writes their records
	int64_t last_window_flagged = 0; // frames of the last in-register window call that were refitted with refinement
	void *frames_buf = nullptr; // scratch of that path (prefix counts, rule counts, records of one slab of frames)
	size_t frames_bytes = 0;
	// Student-t critical values for df = 1..kWindowTcritCap at the confidence level of the last window call
	void *wtab = nullptr;
	size_t wtab_bytes = 0;
	double wtab_conf = -1.0;
	// timing
	bool timing = false;
	std::vector<hipEvent_t> free_events;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> acc_events, solve_events, predict_events;
	// streaming aggregate states created on this context (agg_state.hip); destroying the context releases their
	// device memory and leaves them detached (every later call on them fails, destroy still frees the object)
	std::vector<struct AnofoxHipAggState *> agg_states;
};

namespace anofox {
namespace host {
void agg_state_detach(struct AnofoxHipAggState *s); // agg_state.hip
// host_api.hip: the device batch path (accumulate -> solve -> refine) on device-resident inputs, enqueued on the
// context's stream; used by agg_state.hip to refit the queued groups of a state from its row log
bool refit_groups_device(AnofoxHipContext *ctx, int64_t n_groups, size_t p, int64_t n_rows, const int64_t *d_row_offsets,
                         const double *d_y, const double *const *x_cols, const double *d_w, const AnofoxHipBatchOptions &opt,
                         double *d_core, double *d_inf, AnofoxError *e);
}
}

namespace anofox {
namespace host {


inline void set_error(AnofoxError *e, AnofoxErrorCode code, const std::string &msg) {
	if (!e) return;
	e->code = code;
	const size_t n = msg.size() < 255 ? msg.size() : 255;
	memcpy(e->message, msg.data(), n);
	e->message[n] = 0;
}

inline void reset_error(AnofoxError *e) {
	if (!e) return;
	e->code = ANOFOX_ERROR_SUCCESS;
	memset(e->message, 0, sizeof e->message);
}

inline bool hip_fail(hipError_t rc, const char *what, AnofoxError *e) {
	if (rc == hipSuccess) return false;
	set_error(e, ANOFOX_ERROR_INTERNAL, std::string("HIP error in ") + what + ": " + hipGetErrorString(rc));
	return true;
}

constexpr int kRefineSteps = 2; // iterative-refinement updates applied to queued groups

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline bool ensure_buffer(void **buf, size_t *cap, size_t need, const char *what, AnofoxError *e) {
	if (need <= *cap) return true;
	if (*buf) {
		if (hip_fail(hipFree(*buf), "hipFree", e)) return false; // hipFree synchronises the device
		*buf = nullptr;
		*cap = 0;
	}
	const size_t want = align_up(need + need / 8, 1 << 20);
	if (hipMalloc(buf, want) != hipSuccess) {
		(void)hipGetLastError();
		*buf = nullptr;
		set_error(e, ANOFOX_ERROR_ALLOCATION_FAILURE, std::string("hipMalloc failed for ") + what);
		return false;
	}
	*cap = want;
	return true;
}

inline hipEvent_t get_event(AnofoxHipContext *ctx) {
	if (!ctx->free_events.empty()) {
		hipEvent_t ev = ctx->free_events.back();
		ctx->free_events.pop_back();
		return ev;
	}
	hipEvent_t ev = nullptr;
	(void)hipEventCreate(&ev);
	return ev;
}

} // namespace host
} // namespace anofox
