"""CPU tier: the DuckDB glue (duckdb_shim/fit_agg_hip.cpp), compiled against the stand-in of DuckDB's headers and linked
with the real library, loads without a GPU; registration and bind (option parsing, result type) need none — the device
state is created by the first accepted row."""
import ctypes as C
import os

import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "anofox-statistics_amd", "duckdb_shim", "libanofox_glue_capi.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("libanofox_glue_capi.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(LIB)
    lib.glue_open.restype = C.c_void_p
    lib.glue_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
    lib.glue_close.argtypes = [C.c_void_p]
    lib.glue_result_fields.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("name", ["anofox_stats_ols_fit_agg", "ols_fit_agg", "anofox_stats_ridge_fit_agg", "ridge_fit_agg",
                                  "anofox_stats_wls_fit_agg", "wls_fit_agg"])
def test_every_name_and_alias_binds_with_and_without_options(lib, name):
    msg = C.create_string_buffer(512)
    q = lib.glue_open(name.encode(), None, 0, msg)
    assert q, msg.value
    assert lib.glue_result_fields(q) == 7                      # ols_aggregate.cpp:74-96 without inference
    lib.glue_close(q)
    q = lib.glue_open(name.encode(), b"compute_inference=true;confidence_level=0.9", 0, msg)
    assert q, msg.value
    assert lib.glue_result_fields(q) == 14
    lib.glue_close(q)
    q = lib.glue_open(name.encode(), b"inference=1.0", 1, msg)  # a MAP literal, the alias key, a DOUBLE as boolean
    assert q and lib.glue_result_fields(q) == 14
    lib.glue_close(q)


def test_bad_option_values_fail_at_bind_with_the_reference_texts(lib):
    msg = C.create_string_buffer(512)
    assert not lib.glue_open(b"ols_fit_agg", b"solver=lu", 0, msg)
    assert msg.value.decode() == "Invalid solver: 'lu'. Valid values are 'qr', 'svd', 'cholesky'"   # map_options_parser.cpp:222-234
    assert not lib.glue_open(b"wls_fit_agg", b"hc_type=hc7", 0, msg)
    assert msg.value.decode().startswith("Invalid hc_type: 'hc7'")
    assert not lib.glue_open(b"no_such_agg", None, 0, msg)


FAMILY = [(f"{pre}{m}_{suf}", kind, m) for m in ("ols", "ridge", "wls")
          for pre, suf, kind in (("anofox_stats_", "fit_predict_agg", 0), ("", "fit_predict_agg", 0), ("", "predict_agg", 0),
                                 ("anofox_stats_", "predict_agg", 0), ("anofox_stats_", "fit_predict", 1), ("", "fit_predict", 1))]


@pytest.mark.parametrize("name,kind,model", FAMILY + [("anofox_stats_vif_agg", 2, ""), ("vif_agg", 2, "")])
def test_family_names_aliases_and_overloads_bind(lib, name, kind, model):
    """duckdb_shim/family_agg_hip.cpp: every name the reference registers for the predict aggregates (with the deprecated
    *_predict_agg names, ols_predict_aggregate.cpp:563-602), the window aggregates and vif_agg binds, with every overload, to the
    reference's result type — no GPU needed before the first Finalize."""
    lib.family_open.restype = C.c_void_p
    lib.family_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    lib.family_close.argtypes = [C.c_void_p]
    lib.family_result_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    msg = C.create_string_buffer(512)
    fields = C.c_int(-1)
    overloads = [(None, 0)] if kind == 2 else [(None, 0), (b"fit_intercept=false;null_policy=drop_y_zero_x", 0)]
    if kind == 0:
        overloads += [(None, 1), (b"confidence_level=0.9", 1)]          # (y, x[, weights], split_col[, options])
    for spec, split in overloads:
        q = lib.family_open(name.encode(), spec, 0, split, msg)
        assert q, (name, spec, split, msg.value)
        assert lib.family_result_shape(q, C.byref(fields)) == kind
        assert fields.value == {0: 5, 1: 3, 2: 0}[kind]
        lib.family_close(q)
    if kind != 2:
        assert not lib.family_open(name.encode(), b"null_policy=keep", 0, 0, msg)
        assert msg.value.decode() == "Invalid null_policy: 'keep'. Valid values are 'drop', 'drop_y_zero_x'"   # map_options_parser.cpp:80-93
