// host_sanitize.cpp — ASan / UBSan unit of the library's HOST code (SURVEY.md §5 "sanitizers", CPU tier only:
// GPU AddressSanitizer is not available on this pool).  csrc/host_api.hip and csrc/agg_state.hip are compiled as plain
// C++ by g++ with -fsanitize=address,undefined and linked with this driver; the kernel launchers are stubbed out —
// none is reachable without a GPU: every compute entry point fails with "no HIP device" before it launches anything.
// Exercised here: the reference-compatible symbols' argument checks and error conventions (lib.rs:108-156), the
// validity-bitmask expansion, the host scalar helpers (aic / bic / t_critical / predict_with_interval), the free
// functions (NULL-safe, idempotent), and the argument validation of the batch / window / vif / residual / state
// entry points.  Exit code 0 = every check passed and the sanitizers stayed silent.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../anofox-statistics_amd/csrc/common.h"

namespace anofox {
#define STUB(sig) hipError_t sig { return hipErrorNoDevice; }
STUB(launch_predict(const PredictArgs &, hipStream_t))
STUB(launch_tcrit_table(double *, int, double, hipStream_t))
STUB(launch_window_predict(const WindowArgs &, hipStream_t))
STUB(launch_residuals_narrow(const ResidualArgs &, hipStream_t))
STUB(launch_ingest_chunk(const IngestArgs &, hipStream_t))
STUB(launch_ingest_combine(double *, int64_t *, int64_t, const uint32_t *, const uint32_t *, int64_t, int, int, int, hipStream_t))
STUB(launch_accumulate_wide(const WideArgs &, hipStream_t))
STUB(launch_solve_wide(const WideArgs &, int, hipStream_t))
STUB(launch_solve_tiles(const WideArgs &, hipStream_t))
bool solve_tiles_supports(int) { return false; }
STUB(launch_inference_wide_finish(const WideArgs &, hipStream_t))
STUB(launch_residual_grad_wide(const WideArgs &, hipStream_t))
STUB(launch_accumulate_mid(const WideArgs &, hipStream_t))
STUB(launch_accumulate_quad(const WideArgs &, hipStream_t))
bool accumulate_quad_supports(int, bool, bool, bool) { return false; }
STUB(launch_accumulate_tile(const WideArgs &, hipStream_t))
STUB(launch_accumulate_prefix(const WideArgs &, int, hipStream_t))
bool accumulate_tile_supports(int, bool, bool, bool) { return false; }
STUB(launch_refit_dd_wide(const WideArgs &, hipStream_t))
STUB(launch_refit_dd_narrow(const BatchArgs &, hipStream_t))
STUB(launch_solve_mid(const WideArgs &, int, hipStream_t))
STUB(launch_hc_wide(const WideArgs &, hipStream_t))
STUB(launch_accumulate_narrow(const BatchArgs &, hipStream_t))
STUB(launch_accumulate_narrow_list(const BatchArgs &, const int32_t *, const int32_t *, hipStream_t))
STUB(launch_accumulate_small(const BatchArgs &, int, int32_t *, int32_t *, hipStream_t))
STUB(launch_solve_narrow(const BatchArgs &, hipStream_t))
STUB(launch_refine_fused_narrow(const BatchArgs &, int, hipStream_t))
STUB(launch_vif_narrow(const double *, const int64_t *, int64_t, int, int64_t, double *, hipStream_t))
STUB(launch_vif_from_core(const double *, const int64_t *, int64_t, int, int, int, int64_t, double *, hipStream_t))
STUB(launch_hc_narrow(const BatchArgs &, double *, void *, hipStream_t))
STUB(launch_information_criteria(const double *, int64_t, int, int, int, double *, hipStream_t))
STUB(launch_residuals_wide(const ResidualArgs &, const double *const *, hipStream_t))
STUB(launch_frames_ynn(const double *, int64_t, int64_t *, void *, size_t, hipStream_t))
STUB(launch_frames_from_rows_spec(const int64_t *, int64_t, int64_t, int64_t, int64_t, int64_t *, int64_t *, hipStream_t, const int32_t *, int64_t))
STUB(launch_frames_rule(const FrameArgs &, hipStream_t))
STUB(launch_frames_predict(const FrameArgs &, hipStream_t))
size_t frames_scan_temp_bytes(int64_t) { return 4096; }
STUB(launch_rowlog_sort_slots(const int32_t *, int32_t *, int64_t, void *, size_t, hipStream_t))
STUB(launch_rowlog_iota(int32_t *, int64_t, hipStream_t))
STUB(launch_rowlog_dense(const int32_t *, int64_t, int32_t *, int64_t, hipStream_t))
STUB(launch_rowlog_select(bool, const uint32_t *, const uint8_t *, int64_t, int64_t, const int32_t *, int64_t, unsigned long long *, uint64_t *, unsigned, hipStream_t))
STUB(launch_rowlog_sort_keys(const uint64_t *, uint64_t *, int64_t, unsigned, void *, size_t, hipStream_t))
STUB(launch_rowlog_gather(const uint64_t *, int64_t, int64_t, const RowLogSlab *, int, int, int, double *, double *, size_t, double *, int64_t *, unsigned, hipStream_t))
STUB(launch_rowlog_scatter(const double *, const int32_t *, int64_t, int, double *, const int32_t *, hipStream_t))
STUB(launch_rowlog_positions(const uint32_t *, int64_t, int32_t *, int64_t, hipStream_t))
STUB(launch_rowlog_map_queue(const int32_t *, const int32_t *, const uint32_t *, int32_t *, hipStream_t))
STUB(launch_rowlog_invalidate(uint8_t *, int64_t, const uint32_t *, int64_t, const RowLogSlab *, int, hipStream_t))
STUB(launch_ingest_gather_slots(const double *, const int64_t *, const uint32_t *, int64_t, int, double *, int64_t *, hipStream_t))
STUB(launch_ingest_scatter_slots(double *, int64_t *, const uint32_t *, int64_t, int, const double *, const int64_t *, hipStream_t))
STUB(launch_ingest_clear_slots(double *, int64_t *, const uint32_t *, int64_t, int, hipStream_t))
STUB(launch_rowlog_remap(uint32_t *, int64_t, const uint32_t *, const uint32_t *, int64_t, const RowLogSlab *, int, hipStream_t))
int64_t rowlog_dup_tiles(int64_t rows) { return (rows + 1023) / 1024; }
STUB(launch_rowlog_dup_count(const RowLogSlab *, int, const uint32_t *, const int32_t *, const uint32_t *, int, int64_t *, int64_t, hipStream_t))
STUB(launch_rowlog_dup_fill(const RowLogSlab *, const int64_t *, int, int, int, const uint32_t *, const int32_t *, const uint32_t *, int, const int64_t *,
                            const RowLogSlab &, int64_t, hipStream_t))
size_t rowlog_sort_temp_bytes(int64_t) { return 4096; }
STUB(launch_rowlog_flag_unrefined(const int32_t *, const int32_t *, int64_t, int, double *, double *, hipStream_t))
bool rowlog_key_bits(int64_t, int64_t, unsigned *rb, unsigned *eb) { *rb = 40; *eb = 64; return true; }
size_t ingest_piece_table_bytes(int) { return 4096; }
size_t ingest_sort_temp_bytes(int64_t) { return 4096; }
bool accumulate_mid_supports(int p) { return p > 8 && p <= 32; }
bool solve_mid_supports(int p) { return p > 8 && p <= 32; }
int accumulate_small_segment_width(double) { return 0; }
size_t hc_prep_bytes(int64_t g, int p) { return (size_t)g * (size_t)(p * p + 2 * p + 4) * 8; }
} // namespace anofox

static int failures = 0;
#define CHECK(cond)                                                         \
	do {                                                                    \
		if (!(cond)) {                                                      \
			fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); \
			++failures;                                                     \
		}                                                                   \
	} while (0)

int main() {
	AnofoxError err;
	// ---- scalar helpers (information_criteria.rs:15-85, lib.rs:2217-2349) ----
	double v = 0;
	CHECK(anofox_compute_aic(10.0, 100, 3, &v, &err) && fabs(v - (-224.2585)) < 1e-3);
	CHECK(anofox_compute_bic(10.0, 100, 3, &v, &err) && fabs(v - (-216.4430)) < 1e-3);
	CHECK(anofox_compute_aic(0.0, 10, 2, &v, &err) && isinf(v) && v < 0);
	CHECK(!anofox_compute_aic(1.0, 0, 2, &v, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
	CHECK(!anofox_compute_bic(-1.0, 5, 2, &v, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
	CHECK(!anofox_compute_aic(1.0, 5, 2, nullptr, &err));
	CHECK(fabs(anofox_t_critical(0.95, 10) - 2.2281388519649385) < 1e-9);
	CHECK(fabs(anofox_t_critical(0.95, 1) - 12.706204736174694) < 1e-7);
	CHECK(fabs(anofox_t_critical(0.99, 1000000) - 2.5758326) < 1e-5);
	CHECK(isnan(anofox_t_critical(0.95, 0)) && isnan(anofox_t_critical(1.0, 5)) && isnan(anofox_t_critical(0.0, 5)));
	{
		const double coef[2] = {2.0, NAN}, x[2] = {3.0, 100.0};
		AnofoxPredictionResult r;
		CHECK(anofox_predict_with_interval(coef, 2, 1.0, x, 2, 0.5, 50, 0.95, &r) && r.yhat == 7.0 && r.yhat_lower < 7.0 && r.yhat_upper > 7.0);
		CHECK(anofox_predict_with_interval(coef, 2, NAN, x, 2, NAN, 50, 0.95, &r) && r.yhat == 6.0 && r.yhat_lower == 6.0);
		CHECK(!anofox_predict_with_interval(coef, 2, 1.0, x, 1, 0.5, 50, 0.95, &r) && isnan(r.yhat));
		CHECK(!anofox_predict_with_interval(nullptr, 0, 1.0, x, 0, 0.5, 50, 0.95, &r));
		CHECK(!anofox_predict_with_interval(coef, 2, 1.0, x, 2, 0.5, 50, 0.95, nullptr));
	}
	// ---- free functions: NULL-safe and idempotent (lib.rs:272-311) ----
	anofox_free_result_core(nullptr);
	anofox_free_result_inference(nullptr);
	anofox_free_predictions(nullptr);
	anofox_free_vif(nullptr);
	anofox_free_residuals(nullptr);
	{
		AnofoxFitResultCore c;
		memset(&c, 0, sizeof c);
		c.coefficients = (double *)malloc(3 * sizeof(double));
		anofox_free_result_core(&c);
		anofox_free_result_core(&c);
		CHECK(c.coefficients == nullptr);
		AnofoxFitResultInference f;
		memset(&f, 0, sizeof f);
		f.std_errors = (double *)malloc(8);
		f.ci_upper = (double *)malloc(8);
		anofox_free_result_inference(&f);
		anofox_free_result_inference(&f);
		CHECK(!f.std_errors && !f.ci_upper);
	}
	// ---- the reference-compatible fit symbols: checks made before any GPU work (lib.rs:108-156, ols.rs:38-56) ----
	{
		std::vector<double> y = {1, 2, 3, 4, 5, 6, 7, 8, 9}, x0 = {1, 2, 3, 4, 5, 6, 7, 8, 10}, x1(7, 1.0);
		const uint8_t validity[2] = {0xF7, 0x01}; // row 3 is NULL
		AnofoxDataArray ya = {y.data(), validity, y.size()};
		AnofoxDataArray xa[2] = {{x0.data(), nullptr, x0.size()}, {x1.data(), nullptr, x1.size()}};
		AnofoxOlsOptions oo;
		memset(&oo, 0, sizeof oo);
		oo.fit_intercept = true;
		oo.confidence_level = 0.95;
		AnofoxFitResultCore core;
		AnofoxFitResultInference inf;
		memset(&core, 0xAB, sizeof core);
		CHECK(!anofox_ols_fit(ya, xa, 1, oo, nullptr, nullptr, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
		CHECK(!anofox_ols_fit(ya, nullptr, 1, oo, &core, nullptr, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
		CHECK(!anofox_ols_fit(ya, xa, 0, oo, &core, nullptr, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
		CHECK(!anofox_ols_fit(ya, xa, 2, oo, &core, &inf, &err) && err.code == ANOFOX_ERROR_DIMENSION_MISMATCH);
		AnofoxDataArray empty = {y.data(), nullptr, 0};
		CHECK(!anofox_ols_fit(empty, xa, 1, oo, &core, nullptr, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
		AnofoxRidgeOptions ro;
		memset(&ro, 0, sizeof ro);
		ro.alpha = -1.0;
		CHECK(!anofox_ridge_fit(ya, xa, 1, ro, &core, nullptr, &err) && err.code == ANOFOX_ERROR_INVALID_ALPHA);
		AnofoxWlsOptions wo;
		memset(&wo, 0, sizeof wo);
		AnofoxDataArray wshort = {y.data(), nullptr, 3};
		CHECK(!anofox_wls_fit(ya, xa, 1, wshort, wo, &core, nullptr, &err) && err.code == ANOFOX_ERROR_DIMENSION_MISMATCH);
		// a valid call: inputs are copied and expanded through the validity mask, then the GPU is needed
		const bool ok = anofox_ols_fit(ya, xa, 1, oo, &core, &inf, &err);
		if (!ok) CHECK(err.code == ANOFOX_ERROR_INTERNAL && strstr(err.message, "no HIP device") != nullptr);
		else { anofox_free_result_core(&core); anofox_free_result_inference(&inf); }
		// the same for vif / residuals / predict
		double *out = nullptr;
		size_t out_len = 0;
		CHECK(!anofox_compute_vif(nullptr, 2, &out, &out_len, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
		CHECK(anofox_compute_vif(xa, 1, &out, &out_len, &err) && out_len == 1 && out[0] == 1.0); // vif.rs:30-33
		anofox_free_vif(out);
		CHECK(!anofox_compute_vif(xa, 2, &out, &out_len, &err) && err.code == ANOFOX_ERROR_DIMENSION_MISMATCH);
		AnofoxResidualsResult rr;
		AnofoxDataArray yh = {y.data(), nullptr, 4};
		CHECK(!anofox_compute_residuals(ya, yh, nullptr, 0, NAN, false, &rr, &err) && err.code == ANOFOX_ERROR_DIMENSION_MISMATCH);
		CHECK(!anofox_compute_residuals(empty, empty, nullptr, 0, NAN, false, &rr, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
		CHECK(!anofox_compute_residuals(ya, ya, nullptr, 0, NAN, false, nullptr, &err));
		const double coef[1] = {2.0};
		CHECK(!anofox_predict(xa, 1, coef, 2, 0.0, &out, &out_len, &err) && err.code == ANOFOX_ERROR_DIMENSION_MISMATCH);
		CHECK(!anofox_predict(nullptr, 1, coef, 1, 0.0, &out, &out_len, &err));
	}
	// ---- batch / window / state entry points: argument validation ----
	{
		AnofoxHipContext *ctx = nullptr;
		const bool have = anofox_hip_context_create(-1, &ctx, &err);
		if (!have) CHECK(err.code == ANOFOX_ERROR_INTERNAL && ctx == nullptr);
		CHECK(!anofox_hip_context_create(-1, nullptr, &err));
		AnofoxHipBatchOptions bo;
		memset(&bo, 0, sizeof bo);
		bo.fit_intercept = true;
		bo.confidence_level = 0.95;
		const int64_t off[2] = {0, 3};
		const double col[3] = {1, 2, 4};
		const double *cols[1] = {col};
		double core[7];
		CHECK(!anofox_hip_fit_batch_device(nullptr, 1, 1, 3, off, col, cols, nullptr, bo, core, nullptr, &err) && err.code == ANOFOX_ERROR_INVALID_INPUT);
		AnofoxHipAggState *st = nullptr;
		CHECK(!anofox_hip_agg_state_create(nullptr, 3, bo, 0, &st, &err) && st == nullptr);
		CHECK(!anofox_hip_agg_state_create(ctx, 3, bo, 0, nullptr, &err));
		CHECK(anofox_hip_agg_state_slots(nullptr) == 0 && anofox_hip_agg_state_rows(nullptr) == 0);
		CHECK(!anofox_hip_agg_state_retain_rows(nullptr, 1 << 20, &err));
		CHECK(anofox_hip_agg_state_retaining(nullptr) == 0 && anofox_hip_agg_state_retained_bytes(nullptr) == 0);
		anofox_hip_agg_state_destroy(nullptr);
		CHECK(!anofox_hip_agg_state_update_host(nullptr, 1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, &err));
		CHECK(!anofox_hip_agg_state_finalize_host(nullptr, 0, nullptr, nullptr, nullptr, nullptr, &err));
		CHECK(!anofox_hip_agg_state_combine(nullptr, 1, nullptr, nullptr, &err));
		CHECK(anofox_hip_core_record_len(8) == 14 && anofox_hip_inference_record_len(8) == 42 && anofox_hip_max_features() == 128);
		CHECK(anofox_hip_agg_state_max_features() == 128 && anofox_hip_vif_record_len(5) == 6);
		if (have) anofox_hip_context_destroy(ctx);
		anofox_hip_context_destroy(nullptr);
		anofox_hip_host_free(nullptr);
	}
	if (failures) {
		fprintf(stderr, "%d check(s) failed\n", failures);
		return 1;
	}
	printf("host sanitize unit: all checks passed\n");
	return 0;
}
