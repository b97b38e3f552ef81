/*
 * anofox_stats_hip.h — C ABI of libanofox_stats_hip.so, the MI355X (gfx950)
 * implementation of anofox-statistics' grouped least-squares path
 * (ols_fit_agg / ridge_fit_agg / wls_fit_agg).
 *
 * Two layers:
 *
 *  (1) The reference's own FFI surface for this path, same symbols, same
 *      struct layouts, same ownership and error conventions, so that the
 *      DuckDB C++ layer (src/aggregate_functions/{ols,ridge,wls}_aggregate.cpp,
 *      src/table_functions/{ols,ridge,wls}_fit.cpp, ...) links against this
 *      library instead of the Rust staticlib without source changes.
 *      Each declaration cites the reference interface it replaces
 *      (paths under the reference repository).
 *
 *  (2) New batched entry points (no reference counterpart): one call fits
 *      every group of an aggregate Finalize vector — or a whole GROUP BY —
 *      on the GPU.  Plain pointers and sizes only; no HIP, torch or C++ types.
 *
 * All arithmetic is IEEE-754 binary64.  Every entry point runs on the GPU;
 * there is no CPU fallback: without a usable HIP device the calls fail with
 * ANOFOX_ERROR_INTERNAL and a message.
 */
#ifndef ANOFOX_STATS_HIP_H
#define ANOFOX_STATS_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported symbols (the library is built with -fvisibility=hidden) */
#if defined(__GNUC__)
#define ANOFOX_HIP_API __attribute__((visibility("default")))
#else
#define ANOFOX_HIP_API
#endif

/* ------------------------------------------------------------------------ */
/* (1) Reference-compatible types.  Skipped when the reference's own header */
/* (src/include/anofox_stats_ffi.h) was included first.                     */
/* ------------------------------------------------------------------------ */
#ifndef ANOFOX_STATS_FFI_H

/* replaces AnofoxErrorCode, src/include/anofox_stats_ffi.h:18-31
 * (Rust mirror: crates/anofox-stats-ffi/src/types.rs:6-21) */
typedef enum {
	ANOFOX_ERROR_SUCCESS = 0,
	ANOFOX_ERROR_INVALID_INPUT = 1,
	ANOFOX_ERROR_SINGULAR_MATRIX = 2,
	ANOFOX_ERROR_CONVERGENCE_FAILURE = 3,
	ANOFOX_ERROR_INVALID_ALPHA = 4,
	ANOFOX_ERROR_INVALID_L1_RATIO = 5,
	ANOFOX_ERROR_INSUFFICIENT_DATA = 6,
	ANOFOX_ERROR_ALLOCATION_FAILURE = 7,
	ANOFOX_ERROR_SERIALIZATION_ERROR = 8,
	ANOFOX_ERROR_DIMENSION_MISMATCH = 9,
	ANOFOX_ERROR_NO_VALID_DATA = 10,
	ANOFOX_ERROR_INTERNAL = 99
} AnofoxErrorCode;

/* replaces AnofoxError, anofox_stats_ffi.h:36-39 — 260 bytes, message NUL-terminated, <= 255 chars */
typedef struct {
	AnofoxErrorCode code;
	char message[256];
} AnofoxError;

/* replaces AnofoxDataArray, anofox_stats_ffi.h:44-51 — 24 bytes, passed by value.
 * validity: LSB-first bitmask, bit i == 0 means NULL (treated as NaN, types.rs:66-89); may be NULL. */
typedef struct {
	const double *data;
	const uint8_t *validity;
	size_t len;
} AnofoxDataArray;

/* replaces AnofoxFitResultCore, anofox_stats_ffi.h:56-73 — 64 bytes */
typedef struct {
	double *coefficients; /* malloc'ed by the callee, released by anofox_free_result_core */
	size_t coefficients_len;
	double intercept; /* NaN when fit_intercept == false */
	double r_squared;
	double adj_r_squared;
	double residual_std_error;
	size_t n_observations;
	size_t n_features;
} AnofoxFitResultCore;

/* replaces AnofoxFitResultInference, anofox_stats_ffi.h:78-97 — 72 bytes; arrays are slopes only */
typedef struct {
	double *std_errors;
	double *t_values;
	double *p_values;
	double *ci_lower;
	double *ci_upper;
	size_t len;
	double confidence_level;
	double f_statistic;
	double f_pvalue;
} AnofoxFitResultInference;

/* replaces AnofoxSolverType, anofox_stats_ffi.h:102-106.  Accepted for API
 * compatibility; the GPU path always factors the shifted normal equations by
 * Cholesky (the reference's solvers agree to 1e-10 on well-conditioned data,
 * test/sql/regression/test_map_options.test:65-79). */
typedef enum { ANOFOX_SOLVER_QR = 0, ANOFOX_SOLVER_SVD = 1, ANOFOX_SOLVER_CHOLESKY = 2 } AnofoxSolverType;

/* replaces AnofoxLambdaScaling, anofox_stats_ffi.h:111-114 */
typedef enum { ANOFOX_LAMBDA_SCALING_RAW = 0, ANOFOX_LAMBDA_SCALING_GLMNET = 1 } AnofoxLambdaScaling;

/* replaces AnofoxHcType, anofox_stats_ffi.h:119-125 */
typedef enum {
	ANOFOX_HC_NONE = 0,
	ANOFOX_HC_HC0 = 1,
	ANOFOX_HC_HC1 = 2,
	ANOFOX_HC_HC2 = 3,
	ANOFOX_HC_HC3 = 4
} AnofoxHcType;

/* replaces AnofoxOlsOptions, anofox_stats_ffi.h:130-141 — 24 bytes */
typedef struct {
	bool fit_intercept;
	bool compute_inference;
	double confidence_level;
	AnofoxSolverType solver;
	AnofoxHcType hc_type;
} AnofoxOlsOptions;

/* replaces AnofoxRidgeOptions, anofox_stats_ffi.h:352-364 — 32 bytes */
typedef struct {
	double alpha;
	bool fit_intercept;
	bool compute_inference;
	double confidence_level;
	AnofoxSolverType solver;
	AnofoxLambdaScaling lambda_scaling;
} AnofoxRidgeOptions;

/* replaces AnofoxWlsOptions, anofox_stats_ffi.h:446-457 — 24 bytes */
typedef struct {
	bool fit_intercept;
	bool compute_inference;
	double confidence_level;
	AnofoxSolverType solver;
	AnofoxHcType hc_type;
} AnofoxWlsOptions;

/* replaces anofox_ols_fit, anofox_stats_ffi.h:155-156 (Rust: crates/anofox-stats-ffi/src/lib.rs:98-265).
 * One group, column-major x (one AnofoxDataArray per feature).  Returns false and fills *out_error on
 * failure, handing out no allocation.  out_inference may be NULL; when it is not and the fit produced
 * no inference block it is reset to {NULLs, len 0, NaNs} (types.rs:151-165). */
ANOFOX_HIP_API bool anofox_ols_fit(AnofoxDataArray y, const AnofoxDataArray *x, size_t x_count, AnofoxOlsOptions options,
                    AnofoxFitResultCore *out_core, AnofoxFitResultInference *out_inference, AnofoxError *out_error);

/* replaces anofox_ridge_fit, anofox_stats_ffi.h:378-379 (lib.rs:984-1155) */
ANOFOX_HIP_API bool anofox_ridge_fit(AnofoxDataArray y, const AnofoxDataArray *x, size_t x_count, AnofoxRidgeOptions options,
                      AnofoxFitResultCore *out_core, AnofoxFitResultInference *out_inference, AnofoxError *out_error);

/* replaces anofox_wls_fit, anofox_stats_ffi.h:472-474 (lib.rs:1384-1555) */
ANOFOX_HIP_API bool anofox_wls_fit(AnofoxDataArray y, const AnofoxDataArray *x, size_t x_count, AnofoxDataArray weights,
                    AnofoxWlsOptions options, AnofoxFitResultCore *out_core, AnofoxFitResultInference *out_inference,
                    AnofoxError *out_error);

/* replaces anofox_free_result_core, anofox_stats_ffi.h:161 (lib.rs:272-280) — NULL-safe, idempotent */
ANOFOX_HIP_API void anofox_free_result_core(AnofoxFitResultCore *result);

/* replaces anofox_free_result_inference, anofox_stats_ffi.h:166 (lib.rs:287-311) */
ANOFOX_HIP_API void anofox_free_result_inference(AnofoxFitResultInference *result);

/* replaces anofox_compute_aic / anofox_compute_bic, anofox_stats_ffi.h:570,582 (lib.rs:1932-2011):
 * n ln(rss/n) + 2k  and  n ln(rss/n) + k ln n;  rss == 0 -> -inf;  n == 0 or rss < 0 -> InvalidInput.
 * Scalar helpers, evaluated on the host (they touch no data). */
ANOFOX_HIP_API bool anofox_compute_aic(double rss, size_t n, size_t k, double *out_aic, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_compute_bic(double rss, size_t n, size_t k, double *out_bic, AnofoxError *out_error);

/* ---- prediction helpers of the fit-predict family ---- */

/* replaces anofox_t_critical, anofox_stats_ffi.h:655 (lib.rs:2217-2231): Student-t quantile at (1 + c)/2;
 * NaN when df == 0 or c is outside (0, 1).  Scalar helper, evaluated on the host. */
ANOFOX_HIP_API double anofox_t_critical(double confidence_level, size_t df);

/* replaces AnofoxPredictionResult, anofox_stats_ffi.h:660-667 */
typedef struct {
	double yhat;
	double yhat_lower;
	double yhat_upper;
} AnofoxPredictionResult;

/* replaces anofox_predict_with_interval, anofox_stats_ffi.h:686-688 (lib.rs:2264-2349): one new observation,
 * simplified interval yhat -/+ t * rse * sqrt(1 + 1/n); NaN coefficients are skipped; bounds = yhat when no
 * interval can be formed.  Scalar helper, evaluated on the host. */
ANOFOX_HIP_API bool anofox_predict_with_interval(const double *coefficients, size_t coefficients_len, double intercept,
                                  const double *x_new, size_t x_len, double residual_std_error, size_t n_observations,
                                  double confidence_level, AnofoxPredictionResult *out_result);

/* replaces anofox_predict / anofox_free_predictions, anofox_stats_ffi.h:489-495 (lib.rs:1568-1664,
 * crates/anofox-stats-core/src/models/predict.rs:17-64): y = intercept (0 when NaN) + sum_j coef_j x_j for every row
 * (NaN coefficients propagate, as upstream).  Runs on the GPU; *out_predictions is malloc'ed. */
ANOFOX_HIP_API bool anofox_predict(const AnofoxDataArray *x, size_t x_count, const double *coefficients, size_t coefficients_len,
                    double intercept, double **out_predictions, size_t *out_predictions_len, AnofoxError *out_error);
ANOFOX_HIP_API void anofox_free_predictions(double *predictions);

/* replaces anofox_compute_vif / anofox_free_vif, anofox_stats_ffi.h:516-522 (lib.rs:1688-1750 over
 * crates/anofox-stats-core/src/diagnostics/vif.rs:23-98): feature j regressed on all the others by OLS with an
 * intercept, VIF_j = 1 / (1 - R^2_j); inf when that fit fails or R^2 >= 0.9999, 1 when R^2 < 0; a single feature
 * gives {1.0}.  Runs on the GPU (one Gram matrix for all p regressions); *out_vif is malloc'ed. */
ANOFOX_HIP_API bool anofox_compute_vif(const AnofoxDataArray *x, size_t x_count, double **out_vif, size_t *out_vif_len,
                        AnofoxError *out_error);
ANOFOX_HIP_API void anofox_free_vif(double *vif);

/* replaces AnofoxResidualsResult / anofox_compute_residuals / anofox_free_residuals, anofox_stats_ffi.h:527-558
 * (lib.rs:1752-1920 over crates/anofox-stats-core/src/diagnostics/residuals.rs:30-145): raw = y - y_hat;
 * standardized = raw / s when residual_std_error s is not NaN; with x and include_studentized the leverage
 * h_i = x~_i' (X~'X~)^-1 x~_i of the intercept-augmented design and studentized = raw / (s sqrt(max(1 - h, 1e-10))).
 * A rank-deficient design yields no leverage (has_leverage = false).  Runs on the GPU; x_count <= 32. */
typedef struct {
	double *raw;
	double *standardized;
	double *studentized;
	double *leverage;
	size_t len;
	bool has_standardized;
	bool has_studentized;
	bool has_leverage;
} AnofoxResidualsResult;
ANOFOX_HIP_API bool anofox_compute_residuals(AnofoxDataArray y, AnofoxDataArray y_hat, const AnofoxDataArray *x, size_t x_count,
                              double residual_std_error, bool include_studentized, AnofoxResidualsResult *out_result,
                              AnofoxError *out_error);
ANOFOX_HIP_API void anofox_free_residuals(AnofoxResidualsResult *result);

#endif /* ANOFOX_STATS_FFI_H */

/* ------------------------------------------------------------------------ */
/* (2) Batched GPU entry points (new surface).                               */
/*                                                                          */
/* Natural call site in the reference: the per-state loop of                 */
/* OlsAggFinalize (src/aggregate_functions/ols_aggregate.cpp:249-338, and    */
/* ridge_aggregate.cpp:255-345, wls_aggregate.cpp:268-362), which today      */
/* makes one anofox_*_fit call per group.                                    */
/*                                                                          */
/* Data layout ("grouped columns"): rows sorted by group; group g owns rows  */
/* [row_offsets[g], row_offsets[g+1]).  y is one array of n_rows doubles,    */
/* each feature j is one array x_cols[j] of n_rows doubles (the same         */
/* one-array-per-feature layout as AnofoxDataArray x[] — just concatenated   */
/* over groups), w likewise for WLS.  NULL inputs are encoded as NaN.        */
/* ------------------------------------------------------------------------ */

typedef struct AnofoxHipContext AnofoxHipContext; /* one device + one stream + reusable workspace */

typedef enum { ANOFOX_HIP_MODEL_OLS = 0, ANOFOX_HIP_MODEL_RIDGE = 1, ANOFOX_HIP_MODEL_WLS = 2 } AnofoxHipModel;

/* Union of AnofoxOlsOptions / AnofoxRidgeOptions / AnofoxWlsOptions (same field meaning). */
typedef struct {
	AnofoxHipModel model;
	bool fit_intercept;
	bool compute_inference;
	double confidence_level;
	double alpha; /* ridge only; < 0 -> every group fails with ANOFOX_ERROR_INVALID_ALPHA */
	AnofoxSolverType solver;
	AnofoxLambdaScaling lambda_scaling;
	AnofoxHcType hc_type; /* HC0..HC3 replace se/t/p/ci (OLS, WLS); ignored for ridge and without inference */
} AnofoxHipBatchOptions;

/* Per-group status word stored in the core record: an AnofoxErrorCode, or this value for groups the
 * aggregate maps to SQL NULL before calling the fit (fewer than 2 rows, ols_aggregate.cpp:263-267). */
#define ANOFOX_HIP_STATUS_NULL_TOO_FEW_ROWS 100
/* A streaming aggregate state could not bring this group to the contract's accuracy: its solve asked for the
 * refinement passes (ill-conditioned, or fitting almost exactly) and the group's rows are no longer there — no row log,
 * or one that outgrew its budgets.  The record is NaN; the SQL layer returns NULL and counts it. */
#define ANOFOX_HIP_STATUS_UNREFINED 101

/* Result records, row-major f64, one per group:
 *   core[g]      = { coefficients[0..p), intercept, r_squared, adj_r_squared, residual_std_error,
 *                    n_observations, status }                                   length p + 6
 *   inference[g] = { std_errors[p], t_values[p], p_values[p], ci_lower[p], ci_upper[p],
 *                    f_statistic, f_pvalue }                                    length 5p + 2
 * Fields of the STRUCT the aggregates return (ols_aggregate.cpp:74-96); n_features is the constant p.
 * status != 0  =>  the group is SQL NULL and every other field is NaN.
 * Constant columns give NaN coefficients / inference entries (models/ols.rs:167-171,191-206). */
ANOFOX_HIP_API size_t anofox_hip_core_record_len(size_t n_features);
ANOFOX_HIP_API size_t anofox_hip_inference_record_len(size_t n_features);

/* Largest n_features the library accepts. */
ANOFOX_HIP_API size_t anofox_hip_max_features(void);

/* device_id < 0 selects the current HIP device.  The context owns a stream; calls on one context are
 * serialised, different contexts (e.g. one per DuckDB worker thread) are independent. */
ANOFOX_HIP_API bool anofox_hip_context_create(int device_id, AnofoxHipContext **out_ctx, AnofoxError *out_error);
ANOFOX_HIP_API void anofox_hip_context_destroy(AnofoxHipContext *ctx);

/* Launch on a caller-owned hipStream_t (passed as void*) instead of the context's own (non-blocking) stream.
 * NULL is a stream too — HIP's default stream, which is what PyTorch's default stream is — and is used as given;
 * anofox_hip_context_use_own_stream goes back to the context's stream. */
ANOFOX_HIP_API bool anofox_hip_context_set_stream(AnofoxHipContext *ctx, void *hip_stream, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_context_use_own_stream(AnofoxHipContext *ctx, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_context_synchronize(AnofoxHipContext *ctx, AnofoxError *out_error);
/* Pipelining hook for callers that alternate consecutive batches between two contexts on two streams, so that the
 * small solve / refinement kernels of batch k overlap the HBM-bound accumulate kernel of batch k + 1: the fit entry
 * points make the stream wait for `wait_event` (hipEvent_t as void*, may be NULL) before the accumulate kernel and
 * record `record_event` (may be NULL) right after it.  Chaining the events keeps the accumulate kernels of the two
 * contexts from running against each other.  The gate is ONE-SHOT: it applies to the next fit call on this context
 * only and is cleared by it (the event handles stay the caller's; the library never keeps them past that call). */
ANOFOX_HIP_API bool anofox_hip_context_set_accumulate_gate(AnofoxHipContext *ctx, void *wait_event, void *record_event,
                                            AnofoxError *out_error);

/*
 * Device-resident batch fit.  d_* are device pointers on the context's device; x_cols is a HOST array of
 * n_features device pointers.  d_w may be NULL unless model == WLS.  d_inference may be NULL unless
 * options.compute_inference.  Asynchronous on the context's stream; outputs are complete after
 * anofox_hip_context_synchronize (or a synchronisation of the caller's stream).
 * Returns false (nothing launched) on invalid arguments; per-group failures are reported in the records.
 */
ANOFOX_HIP_API bool anofox_hip_fit_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                 const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                 const double *d_w, AnofoxHipBatchOptions options, double *d_core, double *d_inference,
                                 AnofoxError *out_error);

/* Same with host pointers: stages inputs to the GPU, runs the device path, copies the records back,
 * synchronous.  ctx may be NULL (a per-thread default context on the current device is used). */
ANOFOX_HIP_API bool anofox_hip_fit_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                               const int64_t *row_offsets, const double *y, const double *const *x_cols,
                               const double *w, AnofoxHipBatchOptions options, double *core, double *inference,
                               AnofoxError *out_error);

/*
 * Information criteria as batched outputs of the fit records: what the SQL functions
 * aic(rss, n, k) / bic(rss, n, k) (src/scalar_functions/aic_bic.cpp:12-110 over
 * crates/anofox-stats-core/src/diagnostics/information_criteria.rs:15-33,67-85) give when they are applied to every
 * group's fit with k = estimated parameters (coefficients that are not NaN, + 1 with an intercept) and
 * rss = residual_std_error^2 (n_observations - k):  out[g] = { rss, aic, bic },  aic = n ln(rss / n) + 2 k,
 * bic = n ln(rss / n) + k ln n,  rss == 0 -> -inf (:24-26),  NaN for NULL groups and for n == k.
 * `core` = records of anofox_hip_fit_batch_* fitted with the same `options` (fit_intercept and model are used).
 */
ANOFOX_HIP_API bool anofox_hip_information_criteria_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features,
                                                  const double *d_core, AnofoxHipBatchOptions options, double *d_out,
                                                  AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_information_criteria_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features,
                                                const double *core, AnofoxHipBatchOptions options, double *out,
                                                AnofoxError *out_error);

/*
 * fit + predict in one batch: the `*_fit_predict_agg` aggregates (src/aggregate_functions/ols_predict_aggregate.cpp:
 * 322-425, ridge/wls likewise).  Every row of every group gets {yhat, yhat_lower, yhat_upper} (d_pred, [n_rows x 3],
 * NaN = SQL NULL); rows whose y is NaN (the aggregate's NULL y = "prediction row") or that hold a non-finite
 * feature do not train.  d_train_counts (optional, [n_groups]) is the number of training rows the aggregate's
 * "fewer than 2 -> NULL" rule looks at (ols_predict_aggregate.cpp:333); NULL = use the group's row count.
 * d_core receives the fit records as in anofox_hip_fit_batch_device (compute_inference is ignored: the
 * aggregates fit with inference off).
 */
ANOFOX_HIP_API bool anofox_hip_fit_predict_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                         const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                         const double *d_w, const int64_t *d_train_counts, AnofoxHipBatchOptions options,
                                         double *d_core, double *d_pred, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_fit_predict_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                       const int64_t *row_offsets, const double *y, const double *const *x_cols,
                                       const double *w, const int64_t *train_counts, AnofoxHipBatchOptions options,
                                       double *core, double *pred, AnofoxError *out_error);

/*
 * Expanding-window fit + predict: the `*_fit_predict(y, x [, w] [, opts]) OVER (PARTITION BY .. ORDER BY ..
 * ROWS BETWEEN UNBOUNDED PRECEDING AND CURRENT ROW)` window functions (src/window_functions/ols_fit_predict.cpp:
 * 110-324, ridge_fit_predict.cpp, wls_fit_predict.cpp).  Partitions = groups, rows already in window order.
 * d_pred[3*e .. 3*e+2] = {yhat, yhat_lower, yhat_upper} of x_e from the fit on rows 0..e of e's partition
 * (rows with NaN y — the window's NULL y — or non-finite features do not train); NaN = SQL NULL (at most
 * p + [intercept] training rows so far, failed fit, or non-finite prediction).  A frame ending at 1 PRECEDING is
 * this output shifted down by one row within the partition.  n_features <= 8 run the in-register window kernels,
 * wider designs the virtual-group path (anofox_hip_fit_predict_frames_*).
 */
ANOFOX_HIP_API bool anofox_hip_fit_predict_expanding_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                             const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                             const double *d_w, AnofoxHipBatchOptions options, double *d_pred,
                                             AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_fit_predict_expanding_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                           const int64_t *row_offsets, const double *y, const double *const *x_cols,
                                           const double *w, AnofoxHipBatchOptions options, double *pred,
                                           AnofoxError *out_error);

/*
 * The same window functions over any ROWS frame:
 *   ROWS BETWEEN start_preceding PRECEDING AND end_preceding PRECEDING
 * Offsets count rows before the current row; 0 = CURRENT ROW, negative = FOLLOWING (-3 = 3 FOLLOWING);
 * start_preceding = ANOFOX_HIP_FRAME_UNBOUNDED = UNBOUNDED PRECEDING, end_preceding = -ANOFOX_HIP_FRAME_UNBOUNDED =
 * UNBOUNDED FOLLOWING; start_preceding >= end_preceding.  Frames are clipped to the partition, as DuckDB does.
 * The aggregate trains on the frame's rows with non-NULL y and predicts the x of the LAST row of the frame
 * (ols_fit_predict.cpp:157-162), so with end_preceding = b the output of row e uses x of row e - b (or of the
 * partition's last row when the frame reaches past it); rows whose frame is empty are NULL.  Frames that start
 * UNBOUNDED PRECEDING run the one-pass prefix kernel, the others sum each frame directly (O(frame) per row).
 */
#define ANOFOX_HIP_FRAME_UNBOUNDED INT64_MAX
typedef struct {
	int64_t start_preceding;
	int64_t end_preceding;
} AnofoxHipWindowFrame;
ANOFOX_HIP_API bool anofox_hip_fit_predict_window_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                          const int64_t *d_row_offsets, const double *d_y, const double *const *x_cols,
                                          const double *d_w, AnofoxHipWindowFrame frame, AnofoxHipBatchOptions options,
                                          double *d_pred, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_fit_predict_window_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                        const int64_t *row_offsets, const double *y, const double *const *x_cols,
                                        const double *w, AnofoxHipWindowFrame frame, AnofoxHipBatchOptions options,
                                        double *pred, AnofoxError *out_error);

/*
 * The same window functions over EXPLICIT frames — RANGE and GROUPS frames, EXCLUDE clauses, anything DuckDB's window
 * executor resolves to a row range per output row (src/window_functions/ols_fit_predict.cpp:110-324 receives the
 * frame's rows whatever the frame type): frame of row e = rows [frame_lo[e], frame_hi[e]) of the (already ordered)
 * input, hi exclusive; an empty frame (hi <= lo) is NULL.  Trains on the frame's rows with non-NULL (non-NaN) y,
 * predicts the x of the frame's LAST row hi - 1, NULL unless MORE than p + [intercept] training rows exist (:257-262).
 * Every frame is fitted as a group of the batch path (its refinement passes included), so any n_features <=
 * anofox_hip_max_features() and any frame shape work; cost O(frame) per output row.  The ROWS entry points above use
 * this path for n_features > 8 and for the frames the in-register kernels flag as ill-conditioned.
 */
ANOFOX_HIP_API bool anofox_hip_fit_predict_frames_device(AnofoxHipContext *ctx, int64_t n_rows, size_t n_features, const double *d_y,
                                          const double *const *x_cols, const double *d_w, const int64_t *d_frame_lo,
                                          const int64_t *d_frame_hi, AnofoxHipBatchOptions options, double *d_pred,
                                          AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_fit_predict_frames_host(AnofoxHipContext *ctx, int64_t n_rows, size_t n_features, const double *y,
                                        const double *const *x_cols, const double *w, const int64_t *frame_lo,
                                        const int64_t *frame_hi, AnofoxHipBatchOptions options, double *pred,
                                        AnofoxError *out_error);

/*
 * Grouped variance inflation factors: the Finalize loop of vif_agg (src/aggregate_functions/vif_aggregate.cpp:
 * 144-185) in one call.  x_cols as in the fit entry points (rows of a group contiguous; the aggregate's Update
 * has already dropped NULL rows and NaN values, :67-93).  d_vif[g] = { vif[n_features], status }
 * (anofox_hip_vif_record_len doubles): status 100 = fewer than 3 rows -> SQL NULL (:154), 0 otherwise.
 * n_features <= 8 uses one pass over the rows; up to anofox_hip_vif_max_features() = 129 one grouped fit per feature.
 */
ANOFOX_HIP_API size_t anofox_hip_vif_record_len(size_t n_features);
ANOFOX_HIP_API size_t anofox_hip_vif_max_features(void);
ANOFOX_HIP_API bool anofox_hip_vif_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                 const int64_t *d_row_offsets, const double *const *x_cols, double *d_vif,
                                 AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_vif_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                               const int64_t *row_offsets, const double *const *x_cols, double *vif,
                               AnofoxError *out_error);

/*
 * Grouped residual diagnostics: the Finalize loop of residuals_diagnostics_agg
 * (src/aggregate_functions/residuals_diagnostics_aggregate.cpp:213-286) in one call.  Per row r the record
 * d_out[4 r ..] = { raw, standardized, studentized, leverage } (NaN where the part is absent); per group
 * d_group[2 g ..] = { rows used, ANOFOX_HIP_RESIDUALS_HAS_* flags }.  With drop_nan_rows the rows whose y or
 * y_hat is NaN are left out, as the aggregate's Update does (:154-163; their records are all NaN); the aggregate
 * itself returns NULL for groups with fewer than 3 used rows (:223) and passes no residual standard error (:232),
 * so d_rse (one value per group, NaN = none) may be NULL.  x_cols / n_features may be NULL / 0 (no leverage);
 * n_features <= anofox_hip_residuals_max_features() = 128 (up to 8: one wavefront per group, residuals_narrow.hip;
 * 9 .. 128: one workgroup per group, residuals_wide.hip).
 */
#define ANOFOX_HIP_RESIDUALS_HAS_STANDARDIZED 1
#define ANOFOX_HIP_RESIDUALS_HAS_STUDENTIZED 2
#define ANOFOX_HIP_RESIDUALS_HAS_LEVERAGE 4
ANOFOX_HIP_API size_t anofox_hip_residuals_max_features(void);
ANOFOX_HIP_API bool anofox_hip_residuals_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                       const int64_t *d_row_offsets, const double *d_y, const double *d_y_hat,
                                       const double *const *x_cols, const double *d_rse, bool include_studentized,
                                       bool drop_nan_rows, double *d_out, double *d_group, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_residuals_batch_host(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                     const int64_t *row_offsets, const double *y, const double *y_hat,
                                     const double *const *x_cols, const double *rse, bool include_studentized,
                                     bool drop_nan_rows, double *out, double *group, AnofoxError *out_error);

/*
 * Streaming aggregate state: the Update / Combine / Finalize callbacks of the three aggregates
 * (src/aggregate_functions/ols_aggregate.cpp:120-186,189-234,249-338; ridge_aggregate.cpp:124-345;
 * wls_aggregate.cpp:122-362) with the state kept on the GPU.  The reference buffers every row of every group on
 * the host until Finalize; here ONE object holds an O(p^2) moment record per "slot" (= one DuckDB aggregate state:
 * the shim's Initialize hands out slot numbers 0, 1, 2, ...) and rows are folded in as they arrive, in any order:
 *
 *   update    n_rows rows in arrival order: slot[i] (state of row i), y[i], x_rowmajor[i * p .. i * p + p) (the LIST
 *             child as DuckDB delivers it), w[i] (WLS), valid[i] (optional; 0 = the row Update skips: NULL y, NULL x
 *             list or NULL weight, ols_aggregate.cpp:150-159, wls_aggregate.cpp:160-166).  NULL list ELEMENTS are
 *             passed as NaN (the row filter of the fit drops such rows, ols.rs:59-66).  n_slots = number of slots
 *             handed out so far (every slot[i] < n_slots; the state grows to it).  The feature count is fixed at
 *             creation — the shim keeps the reference's "Inconsistent feature count" check (ols_aggregate.cpp:165-175).
 *             _host: pageable or pinned host memory (anofox_hip_host_alloc), staged through the GPU in chunks with the
 *             copy of one chunk overlapping the kernels of the previous one; returns when the inputs may be reused.
 *             _device: device pointers, asynchronous on the context's stream.
 *   combine   pairs (source slot, target slot) as Combine receives them: the source's rows count as arriving AFTER
 *             the target's (ols_aggregate.cpp:224-233) and the source is emptied; a slot may take part in one pair
 *             per call.
 *   finalize  fit records of slots [0, n_slots) exactly as anofox_hip_fit_batch_* lays them out (status 100 for
 *             fewer than 2 accumulated rows, ols_aggregate.cpp:263-267).  The rows are gone, so the batch path's
 *             refinement passes (re-reading the rows of ill-conditioned or exactly fitting groups) cannot run
 *             unless a row log is kept (below).  A group that would have taken them — smallest Cholesky pivot ratio
 *             below 1e-3 or rss / tss below 1e-7: its coefficients would carry cond^2 eps and its sigma / r^2 the
 *             cancellation of rss = tss - |z|^2 — is NOT handed out as a number: its record is NaN with status
 *             ANOFOX_HIP_STATUS_UNREFINED (101 -> SQL NULL).  *out_unrefined (optional) is the number of such groups
 *             and out_unrefined_slots (optional, room for n_slots entries) receives their slot numbers, in no
 *             particular order.  Every other record meets the batch entry points' tolerances.
 *   retain_rows(max_bytes)  (optional, before the first update) keeps every chunk in a row log in HBM as well — p + 2 (+ 1
 *             with weights) doubles and 5 bytes per row, up to max_bytes — and Finalize then refits exactly the groups
 *             its solve queued, through the batch path (accumulate, solve, refinement passes) on their logged rows:
 *             *out_unrefined is 0 and every group has the batch entry points' accuracy.  Combine re-labels the source
 *             slots' logged rows.  Exceeding max_bytes (or device memory) is not an error: the log continues in
 *             page-locked host memory up to retain_rows_host's max_host_bytes (the kernels of Finalize read those
 *             slabs over PCIe: 5 bytes per row for the selection, the queued groups' rows for the refit); beyond
 *             both budgets the log is dropped, anofox_hip_agg_state_retaining() turns 0 and Finalize flags the queued
 *             groups as without a log.  finalize_device synchronises the stream once when a log is kept.
 *   log-only  Designs of more than 8 features have no moment record to stream into, and HC standard errors (hc_type
 *             other than none, with inference, OLS / WLS) need a second pass over the rows: such a state keeps ONLY the
 *             row log — the reference's row buffers, in HBM — and Finalize runs the batch path over all of it.  Same
 *             entry points and results as above (*out_unrefined is 0); retain_rows(max_bytes) caps the HBM part of the
 *             log (0 = no cap), retain_rows_host its host part, and an Update that exceeds both (or the memory itself)
 *             FAILS with ANOFOX_ERROR_ALLOCATION_FAILURE.
 *
 * A state belongs to one context (device + stream); calls on one state are serialised.  n_features <=
 * anofox_hip_agg_state_max_features() = 128.
 */
typedef struct AnofoxHipAggState AnofoxHipAggState;
ANOFOX_HIP_API size_t anofox_hip_agg_state_max_features(void);
ANOFOX_HIP_API bool anofox_hip_agg_state_create(AnofoxHipContext *ctx, size_t n_features, AnofoxHipBatchOptions options,
                                 int64_t initial_slots, AnofoxHipAggState **out_state, AnofoxError *out_error);
ANOFOX_HIP_API void anofox_hip_agg_state_destroy(AnofoxHipAggState *state);
ANOFOX_HIP_API bool anofox_hip_agg_state_reserve(AnofoxHipAggState *state, int64_t n_slots, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_agg_state_retain_rows(AnofoxHipAggState *state, size_t max_bytes, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_agg_state_retain_rows_host(AnofoxHipAggState *state, size_t max_host_bytes, AnofoxError *out_error);
ANOFOX_HIP_API int anofox_hip_agg_state_retaining(const AnofoxHipAggState *state);          /* 1 while a row log is kept */
ANOFOX_HIP_API size_t anofox_hip_agg_state_retained_bytes(const AnofoxHipAggState *state);  /* HBM held by the log */
ANOFOX_HIP_API size_t anofox_hip_agg_state_retained_host_bytes(const AnofoxHipAggState *state); /* page-locked host memory held by it */
ANOFOX_HIP_API int64_t anofox_hip_agg_state_slots(const AnofoxHipAggState *state);
ANOFOX_HIP_API int64_t anofox_hip_agg_state_rows(const AnofoxHipAggState *state);
ANOFOX_HIP_API bool anofox_hip_agg_state_update_host(AnofoxHipAggState *state, int64_t n_rows, int64_t n_slots, const uint32_t *slot,
                                      const double *y, const double *x_rowmajor, const double *w, const uint8_t *valid,
                                      AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_agg_state_update_device(AnofoxHipAggState *state, int64_t n_rows, int64_t n_slots, const uint32_t *d_slot,
                                        const double *d_y, const double *d_x_rowmajor, const double *d_w, const uint8_t *d_valid,
                                        AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_agg_state_combine(AnofoxHipAggState *state, int64_t n_pairs, const uint32_t *source_slots,
                                  const uint32_t *target_slots, AnofoxError *out_error);
/* combine with preserve_sources = true leaves the sources as they are (DuckDB's AggregateCombineType::PRESERVE_INPUT: a
 * window segment tree combines one node into many frames); the same source may then feed several targets of a call.
 * (r4) A row log — and the rows of a log-only state — follow: every target gets its own copies of its sources' logged rows
 * (the reference's Combine copies the row buffers as well, ols_aggregate.cpp:224-233), appended behind the target's own, so
 * that Finalize refits exactly fitting or ill-conditioned frames as it does for a GROUP BY.  More than 2^24 rows to copy in ONE
 * call is not a window frame's Combine: a moment state then gives its log up (flagged groups), a log-only state reports an error. */
ANOFOX_HIP_API bool anofox_hip_agg_state_combine_ex(AnofoxHipAggState *state, int64_t n_pairs, const uint32_t *source_slots,
                                     const uint32_t *target_slots, bool preserve_sources, AnofoxError *out_error);
/* Finalize of the listed (distinct) slots only: record k belongs to slots[k].  Same records as finalize_host. */
ANOFOX_HIP_API bool anofox_hip_agg_state_finalize_slots_host(AnofoxHipAggState *state, int64_t n_list, const uint32_t *slots, double *core,
                                              double *inference, int64_t *out_unrefined, AnofoxError *out_error);
/* Destroy of aggregate states (ols_aggregate.cpp:108-118): the listed slots become empty and may be handed out again. */
ANOFOX_HIP_API bool anofox_hip_agg_state_release_slots(AnofoxHipAggState *state, int64_t n_list, const uint32_t *slots, AnofoxError *out_error);
/* (r4) Cross-device Combine of moment states (n_features <= 8, no HC errors): the DuckDB glue shards a query's aggregate states
 * over the node's GPUs by hash, and Combine (ols_aggregate.cpp:189-234) may pair a source on one device with a
 * target on another.  export copies the listed slots' records out as the device keeps them — record_len() doubles and the
 * accepted-row count per slot; import overwrites the listed (distinct) slots of ANOTHER state of the same width and options with
 * them, after which an ordinary combine merges them into their targets.  Rows kept in a row log do not travel: both calls give
 * the state's log up, and a group its moments cannot resolve is then flagged by Finalize (status 101), never handed out.
 * Log-only states (wider designs, HC errors) refuse both: they stay on one device. */
/* (r4) Back to the state as created (every slot empty, the row log released, slots() = 0), device buffers kept: between two
 * executions of a prepared statement whose bind data — and with it the query's device state — DuckDB re-uses. */
ANOFOX_HIP_API bool anofox_hip_agg_state_reset(AnofoxHipAggState *state, AnofoxError *out_error);
ANOFOX_HIP_API size_t anofox_hip_agg_state_record_len(const AnofoxHipAggState *state);
ANOFOX_HIP_API bool anofox_hip_agg_state_export_slots_host(AnofoxHipAggState *state, int64_t n_list, const uint32_t *slots, double *records,
                                            int64_t *counts, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_agg_state_import_slots_host(AnofoxHipAggState *state, int64_t n_list, const uint32_t *slots, const double *records,
                                            const int64_t *counts, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_agg_state_finalize_host(AnofoxHipAggState *state, int64_t n_slots, double *core, double *inference,
                                        int64_t *out_unrefined, int32_t *out_unrefined_slots, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_agg_state_finalize_device(AnofoxHipAggState *state, int64_t n_slots, double *d_core, double *d_inference,
                                          AnofoxError *out_error);
/* Page-locked host memory for the shim's row arenas (copies from it run at the full PCIe rate and asynchronously). */
ANOFOX_HIP_API void *anofox_hip_host_alloc(size_t bytes);
ANOFOX_HIP_API void anofox_hip_host_free(void *ptr);

/* Predictions only, from existing fit records (d_core as produced by the fit entry points). */
ANOFOX_HIP_API bool anofox_hip_predict_batch_device(AnofoxHipContext *ctx, int64_t n_groups, size_t n_features, int64_t n_rows,
                                     const int64_t *d_row_offsets, const double *const *x_cols, const double *d_core,
                                     double confidence_level, double *d_pred, AnofoxError *out_error);

/*
 * Multi-GPU (no reference counterpart): the path shards by GROUP BY key — one process per GPU fits its
 * own keys with no data-path communication — and ONE collective assembles the result: an RCCL all-gather of the
 * fixed-size f64 records over xGMI.  These entry points are that exchange step without Python: rank 0 calls
 * anofox_hip_comm_unique_id and ships the 128 bytes to the other ranks by whatever channel the host has; every rank
 * calls anofox_hip_comm_create on its own context (collective: returns when all ranks have joined); after a fit,
 * anofox_hip_gather_records_device gathers `records_per_rank` records of `record_len` doubles from every rank into
 * d_all[world_size * records_per_rank * record_len] on every rank, rank r's block at r * records_per_rank (shards are
 * padded to the same count; pad records with NaN), stream-ordered behind the fit kernels of the communicator's
 * context.  RCCL is loaded at run time by the first of these calls (dlopen): the library has no link-time dependency
 * on it.
 */
#define ANOFOX_HIP_COMM_ID_BYTES 128
typedef struct AnofoxHipComm AnofoxHipComm;
ANOFOX_HIP_API bool anofox_hip_comm_unique_id(uint8_t *out_id /* [ANOFOX_HIP_COMM_ID_BYTES] */, AnofoxError *out_error);
ANOFOX_HIP_API bool anofox_hip_comm_create(AnofoxHipContext *ctx, int world_size, int rank, const uint8_t *id, AnofoxHipComm **out_comm,
                            AnofoxError *out_error);
ANOFOX_HIP_API void anofox_hip_comm_destroy(AnofoxHipComm *comm);
ANOFOX_HIP_API int anofox_hip_comm_world_size(const AnofoxHipComm *comm);
ANOFOX_HIP_API int anofox_hip_comm_rank(const AnofoxHipComm *comm);
/* ranks RCCL itself counts in the communicator (ncclCommCount): the audit figure of an N-GPU run; 0 = unknown */
ANOFOX_HIP_API int anofox_hip_comm_ranks_seen(const AnofoxHipComm *comm);
ANOFOX_HIP_API bool anofox_hip_gather_records_device(AnofoxHipComm *comm, const double *d_local, int64_t records_per_rank,
                                      size_t record_len, double *d_all, AnofoxError *out_error);

/* Measurement hooks (bench.py): when enabled, every accumulate-kernel launch of this context is
 * bracketed by HIP events on the launch stream. */
typedef struct {
	double accumulate_ms;     /* summed duration of the dominant (HBM-streaming) kernel */
	int64_t accumulate_count; /* launches summed */
	double solve_ms;          /* summed duration of the per-group solve/diagnostics kernels */
	int64_t solve_count;
	double predict_ms;        /* summed duration of the per-row prediction kernel */
	int64_t predict_count;
	double accumulate_ms_min; /* (r4) shortest / longest single launch of the dominant kernel in the interval (0 if none) */
	double accumulate_ms_max;
} AnofoxHipKernelTimes;
ANOFOX_HIP_API bool anofox_hip_context_enable_timing(AnofoxHipContext *ctx, bool enable, AnofoxError *out_error);
/* synchronises, returns the sums since the last call and resets them */
ANOFOX_HIP_API bool anofox_hip_context_collect_timing(AnofoxHipContext *ctx, AnofoxHipKernelTimes *out, AnofoxError *out_error);

/* Diagnostic: how many groups of the context's most recent fit launch (narrow path: the whole batch; wide path:
 * the last slab) were queued for the on-device refinement passes.  Synchronises the context's stream. */
ANOFOX_HIP_API bool anofox_hip_context_last_refine_count(AnofoxHipContext *ctx, int64_t *out_count, AnofoxError *out_error);

/* Diagnostic: output rows of the context's most recent window call (n_features <= 8) whose frame the in-register
 * kernel flagged as ill-conditioned and that were refitted through the virtual-group path with refinement. */
ANOFOX_HIP_API bool anofox_hip_context_last_window_refit_count(AnofoxHipContext *ctx, int64_t *out_count, AnofoxError *out_error);

/* Library / build identification, e.g. "anofox_stats_hip 0.1 gfx950". */
ANOFOX_HIP_API const char *anofox_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ANOFOX_STATS_HIP_H */
