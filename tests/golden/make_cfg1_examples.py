#!/usr/bin/env python3
"""Writes the `ols_single_series_examples` block of tests/golden/known_answers.json.

BASELINE config 1 = examples/ols_single_series.sql of the reference: eight example calls of anofox_stats_ols_fit
(inputs transcribed below with their line ranges; data only).  The example file holds NO expected output (it is a
demo script, and its comments state only "slope=5, intercept=5" for example 1), so — as SURVEY.md §8d prescribes for
cfg1 — the expected values are the closed form (numpy least squares + scipy distributions), rounded as the example's
own SELECT list rounds them.  Example 7 draws its noise from RANDOM(); it is replayed with the noise-free line.
Run from the repo root:  python tests/golden/make_cfg1_examples.py
"""
import json
import os

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))

EXAMPLES = [
    dict(name="ex1_basic_fit", source="examples/ols_single_series.sql:21-32", y=[10, 15, 20, 25, 30], x=[[1, 2, 3, 4, 5]],
         options={"intercept": True}, project={"coefficients": None, "intercept": None, "r_squared": 4}),
    dict(name="ex2_multiple_regression", source="examples/ols_single_series.sql:40-57",
         y=[15, 22, 31, 38, 45, 54, 61, 70],
         x=[[1, 2, 3, 4, 5, 6, 7, 8], [2, 3, 5, 6, 7, 9, 10, 12], [3, 4, 4, 5, 6, 6, 7, 7]],
         options={"intercept": True},
         # the example's design is exactly rank deficient (x2 = x1 + x3 - 2): which column is dropped / min-norm is
         # unpinned upstream, so only the projection-invariant outputs are expected
         project={"r_squared": 4, "n_observations": None, "n_features": None}, note="rank deficient: x2 = x1 + x3 - 2"),
    dict(name="ex3_full_inference", source="examples/ols_single_series.sql:65-82",
         y=[10, 15, 20, 25, 30, 35, 40, 45, 50, 55], x=[[1, 2, 3, 4, 5, 6, 7, 8, 9, 10]],
         options={"intercept": True, "compute_inference": True, "confidence_level": 0.95},
         project={"coefficients": None, "intercept": None}),
    dict(name="ex4_diagnostics", source="examples/ols_single_series.sql:90-102",
         y=[12.5, 17.2, 21.8, 26.1, 31.5, 35.9, 41.2, 45.8], x=[[1, 2, 3, 4, 5, 6, 7, 8]], options={"intercept": True},
         project={"r_squared": 4, "adj_r_squared": 4, "residual_std_error": 4, "n_observations": None, "n_features": None}),
    dict(name="ex5_prediction", source="examples/ols_single_series.sql:111-125", y=[10, 20, 30, 40, 50], x=[[1, 2, 3, 4, 5]],
         options={"intercept": True}, project={"coefficients": None, "intercept": None}, predict_x=[6, 7, 8, 9, 10], predict_round=2),
    dict(name="ex6_with_intercept", source="examples/ols_single_series.sql:134-145", y=[5, 10, 15, 20, 25], x=[[1, 2, 3, 4, 5]],
         options={"intercept": True}, project={"intercept": 4, "coefficients": 4, "r_squared": 4}),
    dict(name="ex6_without_intercept", source="examples/ols_single_series.sql:146-158", y=[5, 10, 15, 20, 25], x=[[1, 2, 3, 4, 5]],
         options={"intercept": False}, project={"coefficients": 4, "r_squared": 4}),
    dict(name="ex7_fit_from_table_noise_free", source="examples/ols_single_series.sql:166-187",
         y=[2.5 * i + 10.0 for i in range(1, 21)], x=[[float(i) for i in range(1, 21)]], options={"intercept": True},
         project={"intercept": 2, "coefficients": 2, "r_squared": 4, "n_observations": None}),
] + [
    dict(name=f"ex8_confidence_{int(c * 100)}", source=f"examples/ols_single_series.sql:{lo}-{hi}",
         y=[10, 20, 30, 40, 50, 60, 70, 80], x=[[1, 2, 3, 4, 5, 6, 7, 8]],
         options={"intercept": True, "compute_inference": True, "confidence_level": c}, project={"ci_lower": 4, "ci_upper": 4})
    for c, lo, hi in ((0.90, 195, 206), (0.95, 208, 219), (0.99, 221, 232))
]


def closed_form(y, x, opts):
    y = np.asarray(y, dtype=np.float64)
    X = np.stack([np.asarray(c, dtype=np.float64) for c in x], axis=1)
    n, p = X.shape
    icpt = bool(opts.get("intercept", True))
    A = np.hstack([np.ones((n, 1)), X]) if icpt else X
    beta = np.linalg.lstsq(A, y, rcond=None)[0]
    r = y - A @ beta
    rss = float(r @ r)
    tss = float(((y - y.mean()) ** 2).sum()) if icpt else float(y @ y)
    pp = p + int(icpt)
    df = n - pp
    out = {"coefficients": beta[int(icpt):].tolist(), "intercept": float(beta[0]) if icpt else None,
           "r_squared": 1.0 - rss / tss, "adj_r_squared": 1.0 - (rss / tss) * (n - int(icpt)) / df,
           "residual_std_error": float(np.sqrt(rss / df)), "n_observations": n, "n_features": p}
    if opts.get("compute_inference"):
        cov = rss / df * np.linalg.inv(A.T @ A)
        se = np.sqrt(np.diag(cov))[int(icpt):]
        tq = stats.t.ppf(0.5 * (1 + opts.get("confidence_level", 0.95)), df)
        b = np.asarray(out["coefficients"])
        out["ci_lower"], out["ci_upper"] = (b - tq * se).tolist(), (b + tq * se).tolist()
    return out


def main():
    block = []
    for ex in EXAMPLES:
        cf = closed_form(ex["y"], ex["x"], ex["options"])
        expect = {}
        for key, rnd in ex["project"].items():
            v = cf[key]
            if isinstance(v, list):
                expect[key] = [[round(t, rnd if rnd is not None else 9) + 0.0, rnd if rnd is not None else 9] for t in v]
            elif isinstance(v, int):
                expect[key] = v
            elif v is not None:
                expect[key] = [round(v, rnd if rnd is not None else 9) + 0.0, rnd if rnd is not None else 9]
        e = {"name": ex["name"], "source": ex["source"], "y": [float(v) for v in ex["y"]],
             "x": [[float(v) for v in c] for c in ex["x"]], "options": ex["options"], "expect": expect}
        if "note" in ex:
            e["note"] = ex["note"]
        if "predict_x" in ex:
            e["predict_x"] = ex["predict_x"]
            e["predict_expect"] = [round(cf["intercept"] + cf["coefficients"][0] * xv, ex["predict_round"]) for xv in ex["predict_x"]]
        block.append(e)
    path = os.path.join(HERE, "known_answers.json")
    d = json.load(open(path))
    d["ols_single_series_examples"] = {
        "_comment": "BASELINE cfg1: the eight example calls of examples/ols_single_series.sql (inputs transcribed, data only). "
                    "The example file holds no expected output; 'expect' is the closed form rounded as the example's SELECT "
                    "list rounds ([value, decimals]; 9 decimals where it does not round) — written by tests/golden/make_cfg1_examples.py.",
        "cases": block}
    json.dump(d, open(path, "w"), indent=1)
    print(f"{len(block)} example calls written")


if __name__ == "__main__":
    main()
