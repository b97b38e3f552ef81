"""Options surface of the aggregates (reference tests
test/sql/regression/test_map_options.test:47-85,91-131,137-216,320-345)."""
import pytest

from conftest import import_pkg


def test_defaults_match_bind_data():
    pkg = import_pkg()
    o = pkg.parse_options(None)
    assert (o.fit_intercept, o.compute_inference, o.confidence_level) == (True, False, 0.95)
    assert (o.solver, o.hc_type, o.lambda_scaling, o.alpha) == ("svd", "none", "raw", 1.0)


def test_aliases_case_insensitivity_and_unknown_keys():
    pkg = import_pkg()
    o = pkg.parse_options({"INTERCEPT": False, "Inference": 1, "confidence": 0.9, "full_output": True, "Solver": "QR"})
    assert not o.fit_intercept and o.compute_inference and o.confidence_level == 0.9 and o.solver == "qr"
    o = pkg.parse_options({"fit_intercept": 0.0, "compute_inference": 2})
    assert not o.fit_intercept and o.compute_inference
    assert pkg.parse_options({"lambda": 0.3}).alpha == 0.3
    assert pkg.parse_options({"lambda": 0.3, "alpha": 0.7}).alpha == 0.7      # alpha wins
    assert pkg.parse_options({"hc_type": "HC3", "lambda_scaling": "GLMNET"}).hc_type == "hc3"
    assert pkg.parse_options({"confidence_level": 1.7}).confidence_level == 1.7   # no range check in the reference
    assert pkg.parse_options({"solver": None}).solver == "svd"


@pytest.mark.parametrize("opts,msg", [
    ({"solver": "lu"}, "Invalid solver: 'lu'. Valid values are 'qr', 'svd', 'cholesky'"),
    ({"hc_type": "hc9"}, "Invalid hc_type: 'hc9'. Valid values are 'none', 'hc0', 'hc1', 'hc2', 'hc3'"),
    ({"lambda_scaling": "sklearn"}, "Invalid lambda_scaling: 'sklearn'. Valid values are 'raw', 'glmnet'"),
    ({"intercept": "yes"}, "Cannot convert value of type STR to boolean"),
])
def test_error_messages(opts, msg):
    pkg = import_pkg()
    with pytest.raises(pkg.InvalidInputException) as ei:
        pkg.parse_options(opts)
    assert msg in str(ei.value)


def test_batch_options_struct():
    pkg = import_pkg()
    b = pkg.parse_options({"alpha": 2.0, "lambda_scaling": "glmnet", "inference": True}).batch_options("ridge")
    assert (b.model, b.alpha, b.lambda_scaling, b.compute_inference, b.fit_intercept) == (1, 2.0, 1, True, True)


def test_sql_function_names_registered():
    pkg = import_pkg()
    for name in ("anofox_stats_ols_fit_agg", "ols_fit_agg", "anofox_stats_ridge_fit_agg", "ridge_fit_agg",
                 "anofox_stats_wls_fit_agg", "wls_fit_agg"):
        assert name in pkg.SQL_FUNCTIONS


def test_sql_function_table_covers_the_reference_registrations():
    """Every name the reference registers for this path (ScalarFunctionSet / AggregateFunctionSet names in
    src/{aggregate,window,table,scalar}_functions for ols / ridge / wls / vif / aic / bic / predict)."""
    pkg = import_pkg()
    names = """aic anofox_stats_aic anofox_stats_bic anofox_stats_ols_fit anofox_stats_ols_fit_agg
        anofox_stats_ols_fit_predict anofox_stats_ols_fit_predict_agg anofox_stats_predict anofox_stats_ridge_fit
        anofox_stats_ridge_fit_agg anofox_stats_ridge_fit_predict anofox_stats_ridge_fit_predict_agg anofox_stats_vif
        anofox_stats_vif_agg anofox_stats_wls_fit anofox_stats_wls_fit_agg anofox_stats_wls_fit_predict
        anofox_stats_wls_fit_predict_agg bic ols_fit ols_fit_agg ols_fit_predict ols_fit_predict_agg ols_predict_agg
        ridge_fit ridge_fit_agg ridge_fit_predict ridge_fit_predict_agg ridge_predict_agg vif vif_agg wls_fit wls_fit_agg
        wls_fit_predict wls_fit_predict_agg wls_predict_agg""".split()
    missing = [n for n in names if n not in pkg.SQL_FUNCTIONS]
    assert not missing, missing
