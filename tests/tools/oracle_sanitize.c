/* oracle_sanitize.c — ASan / UBSan run of the CPU oracle (oracle/anofox_oracle.c) on the shapes the parity tests
 * feed it: ragged groups, empty and one-row groups, NaN / inf rows, non-positive weights, constant and collinear
 * columns, every model and option, several threads; plus the fit-predict, window, VIF and residual entry points.
 * Test infrastructure (CPU tier): exit code 0 = the sanitizers stayed silent and the outputs are sane. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	int32_t model, fit_intercept, compute_inference, lambda_scaling;
	double confidence_level, alpha;
	int32_t hc_type, plain_qr, plain_svd;
} OracleOptions; /* (oracle/anofox_oracle.c:65-79) */

int oracle_fit_groups(const double *y, const double *const *x, const double *w, const int64_t *offsets, int64_t n_groups, size_t p,
                      const OracleOptions *opt, double *core, double *inf, int n_threads);
int oracle_fit_predict_groups(const double *y, const double *const *x, const double *w, const int64_t *offsets, int64_t n_groups,
                              size_t p, const OracleOptions *opt, const int64_t *train_counts, double *core, double *pred);
int oracle_fit_predict_window(const double *y, const double *const *x, const double *w, const int64_t *offsets, int64_t n_groups,
                              size_t p, const OracleOptions *opt, int64_t start_preceding, int64_t end_preceding, double *pred);
int oracle_vif_groups(const double *const *x, const int64_t *offsets, int64_t n_groups, size_t p, int64_t min_rows, double *out);
int oracle_residuals_groups(const double *y, const double *y_hat, const double *const *x, const int64_t *offsets, int64_t n_groups,
                            size_t p, const double *rse, int include_studentized, int drop_nan_rows, double *out, double *group);
double oracle_t_critical(double confidence_level, int64_t df);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}

int main(void) {
	enum { G = 40, P = 6 };
	int64_t offs[G + 1];
	offs[0] = 0;
	for (int g = 0; g < G; ++g) {
		int n = (g == 3) ? 0 : (g == 4) ? 1 : (g == 5) ? 2 : (g == 6) ? P + 1 : 5 + (int)(urand() * 200);
		offs[g + 1] = offs[g] + n;
	}
	const size_t N = (size_t)offs[G];
	double *y = malloc(N * sizeof *y), *w = malloc(N * sizeof *w), *yhat = malloc(N * sizeof *yhat);
	double *cols[P];
	for (int j = 0; j < P; ++j) cols[j] = malloc(N * sizeof(double));
	for (size_t i = 0; i < N; ++i) {
		double acc = 1.0;
		for (int j = 0; j < P; ++j) {
			cols[j][i] = urand() * 20.0 - 10.0;
			acc += (j + 1) * cols[j][i];
		}
		y[i] = acc + urand();
		w[i] = 0.5 + urand();
		yhat[i] = acc;
		if (urand() < 0.02) y[i] = NAN;
		if (urand() < 0.02) cols[2][i] = INFINITY;
		if (urand() < 0.02) w[i] = -1.0;
	}
	for (int64_t i = offs[8]; i < offs[9]; ++i) cols[1][i] = 3.25;                 /* constant column */
	for (int64_t i = offs[9]; i < offs[10]; ++i) cols[3][i] = 2.0 * cols[0][i] + 1; /* collinear */
	for (int64_t i = offs[10]; i < offs[11]; ++i) y[i] = NAN;                       /* no valid row */
	double *core = malloc(G * (P + 6) * sizeof(double)), *inf = malloc(G * (5 * P + 2) * sizeof(double));
	double *pred = malloc(N * 3 * sizeof(double)), *vif = malloc(G * (P + 1) * sizeof(double));
	double *res = malloc(N * 4 * sizeof(double)), *grp = malloc(G * 2 * sizeof(double));
	int bad = 0, runs = 0;
	for (int model = 0; model < 3; ++model)
		for (int icpt = 0; icpt < 2; ++icpt)
			for (int infr = 0; infr < 2; ++infr)
				for (int hc = 0; hc <= 4; hc += 2)
					for (int plain = 0; plain < 3; ++plain) { /* refined (the checker), plain QR, plain SVD */
						OracleOptions o;
						memset(&o, 0, sizeof o);
						o.model = model; o.fit_intercept = icpt; o.compute_inference = infr; o.lambda_scaling = plain & 1;
						o.confidence_level = 0.9; o.alpha = 0.7; o.hc_type = hc; o.plain_qr = plain == 1; o.plain_svd = plain == 2;
						oracle_fit_groups(y, (const double *const *)cols, model == 2 ? w : NULL, offs, G, P, &o, core, infr ? inf : NULL, 1 + runs % 4);
						++runs;
						/* a healthy group has status 0 and a finite r^2 in [0, 1] (uncentred without intercept) */
						const double *c = core + 20 * (P + 6);
						if (c[P + 5] != 0.0 || !(c[P + 1] > 0.5 && c[P + 1] <= 1.0 + 1e-12)) ++bad;
						if (core[3 * (P + 6) + P + 5] != 100.0 || core[10 * (P + 6) + P + 5] != 10.0) ++bad;
					}
	OracleOptions o;
	memset(&o, 0, sizeof o);
	o.fit_intercept = 1; o.confidence_level = 0.95; o.alpha = 1.0;
	oracle_fit_predict_groups(y, (const double *const *)cols, NULL, offs, G, P, &o, NULL, core, pred);
	oracle_fit_predict_window(y, (const double *const *)cols, NULL, offs, G, 2, &o, INT64_MAX, 0, pred);
	oracle_fit_predict_window(y, (const double *const *)cols, NULL, offs, G, 2, &o, 7, -2, pred);
	oracle_vif_groups((const double *const *)cols, offs, G, P, 3, vif);
	oracle_residuals_groups(y, yhat, (const double *const *)cols, offs, G, P, NULL, 1, 1, res, grp);
	if (!(fabs(oracle_t_critical(0.95, 10) - 2.2281388519649385) < 1e-9)) ++bad;
	free(y); free(w); free(yhat); free(core); free(inf); free(pred); free(vif); free(res); free(grp);
	for (int j = 0; j < P; ++j) free(cols[j]);
	if (bad) { fprintf(stderr, "%d sanity check(s) failed over %d runs\n", bad, runs); return 1; }
	printf("oracle sanitize unit: %d option combinations, all clean\n", runs);
	return 0;
}
