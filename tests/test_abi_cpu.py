"""CPU-side checks of the drop-in boundary: the header compiles as C, struct layouts match the reference's
ABI, the shared library loads and exports every declared symbol, and — with no GPU —
the entry points fail loudly instead of falling back to the CPU."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT, import_pkg

HEADER = os.path.join(ROOT, "include", "anofox_stats_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"ANOFOX_HIP_API[^;(]*?\b(anofox_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_surface():
    names = _declared_functions()
    for must in ("anofox_ols_fit", "anofox_ridge_fit", "anofox_wls_fit", "anofox_free_result_core",
                 "anofox_free_result_inference", "anofox_compute_aic", "anofox_compute_bic",
                 "anofox_hip_fit_batch_device", "anofox_hip_fit_batch_host", "anofox_hip_context_create"):
        assert must in names


def test_library_exports_every_declared_symbol():
    abi = import_pkg("_abi")
    lib = abi.load()
    declared = _declared_functions()
    assert sorted(abi.SYMBOLS) == declared          # the ctypes table and the header agree
    for name in declared:
        assert getattr(lib, name) is not None


def test_header_compiles_as_c_and_layouts_match_reference_abi():
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "anofox_stats_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(AnofoxError), sizeof(AnofoxDataArray),
         sizeof(AnofoxFitResultCore), sizeof(AnofoxFitResultInference), sizeof(AnofoxOlsOptions),
         sizeof(AnofoxRidgeOptions), sizeof(AnofoxWlsOptions), sizeof(AnofoxErrorCode), sizeof(AnofoxHipBatchOptions));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(AnofoxError, message), offsetof(AnofoxDataArray, len),
         offsetof(AnofoxFitResultCore, n_features), offsetof(AnofoxFitResultInference, f_pvalue),
         offsetof(AnofoxOlsOptions, hc_type), offsetof(AnofoxRidgeOptions, lambda_scaling));
  return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    sizes = [int(v) for v in out[0].split()]
    offs = [int(v) for v in out[1].split()]
    # SURVEY.md Appendix A (measured against src/include/anofox_stats_ffi.h with gcc 11.4, x86-64 SysV)
    assert sizes[:8] == [260, 24, 64, 72, 24, 32, 24, 4]
    assert offs == [4, 16, 56, 64, 20, 28]
    abi = import_pkg("_abi")
    assert C.sizeof(abi.AnofoxError) == 260 and C.sizeof(abi.AnofoxDataArray) == 24
    assert C.sizeof(abi.AnofoxFitResultCore) == 64 and C.sizeof(abi.AnofoxFitResultInference) == 72
    assert C.sizeof(abi.AnofoxOlsOptions) == 24 and C.sizeof(abi.AnofoxRidgeOptions) == 32
    assert C.sizeof(abi.AnofoxHipBatchOptions) == sizes[8]


def test_header_coexists_with_reference_guard():
    # when the reference's own header came first (its include guard is defined) only the batch API is added
    prog = '#define ANOFOX_STATS_FFI_H\n#include <stdbool.h>\n#include <stddef.h>\n#include <stdint.h>\n' \
           'typedef struct { int code; char message[256]; } AnofoxError;\n' \
           'typedef enum { Q = 0 } AnofoxSolverType; typedef enum { R = 0 } AnofoxLambdaScaling; ' \
           'typedef enum { H = 0 } AnofoxHcType;\n#include "anofox_stats_hip.h"\nint main(void){return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", os.path.join(d, "t.o")])


def test_helpers_that_need_no_gpu():
    pkg = import_pkg()
    abi = import_pkg("_abi")
    lib = abi.load()
    assert lib.anofox_hip_core_record_len(8) == 14 and lib.anofox_hip_inference_record_len(8) == 42
    assert lib.anofox_hip_max_features() >= 8
    assert b"gfx950" in lib.anofox_hip_version()
    assert abs(pkg.aic(10.0, 100, 3) - (-224.2585)) < 1e-3       # information_criteria.rs:110-123
    assert abs(pkg.bic(10.0, 100, 3) - (-216.4430)) < 1e-3
    assert pkg.aic(0.0, 10, 2) == float("-inf")
    assert pkg.aic(1.0, 0, 2) is None and pkg.bic(-1.0, 10, 2) is None     # error -> SQL NULL
    core = abi.AnofoxFitResultCore()
    lib.anofox_free_result_core(C.byref(core))                              # NULL-safe on zeroed structs
    lib.anofox_free_result_core(None)
    inf = abi.AnofoxFitResultInference()
    lib.anofox_free_result_inference(C.byref(inf))
    lib.anofox_free_result_inference(None)


def test_no_silent_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pkg = import_pkg()
    with pytest.raises(pkg.AnofoxStatsError, match="no HIP device"):
        pkg.Context()
    with pytest.raises(pkg.InvalidInputException, match="no HIP device"):
        pkg.ols_fit([1.0, 2.0, 3.5], [[1.0, 2.0, 3.0]])
    with pytest.raises(pkg.AnofoxStatsError, match="no HIP device"):
        pkg.ols_fit_agg([0, 0, 0], [1.0, 2.0, 3.5], [[1.0], [2.0], [3.0]])


def test_argument_validation_precedes_device_use():
    """NULL out_core / empty x are rejected with InvalidInput before anything touches the GPU (lib.rs:113-125)."""
    abi = import_pkg("_abi")
    lib = abi.load()
    err = abi.AnofoxError()
    y = abi.AnofoxDataArray()
    opt = abi.AnofoxOlsOptions(True, False, 0.95, 1, 0)
    assert not lib.anofox_ols_fit(y, None, 0, opt, None, None, C.byref(err))
    assert err.code == abi.ERROR_INVALID_INPUT and err.text() == "out_core is NULL"
    core = abi.AnofoxFitResultCore()
    assert not lib.anofox_ols_fit(y, None, 0, opt, C.byref(core), None, C.byref(err))
    assert err.code == abi.ERROR_INVALID_INPUT and err.text() == "x is NULL or empty"
    ropt = abi.AnofoxRidgeOptions(-1.0, True, False, 0.95, 1, 0)
    xs = (abi.AnofoxDataArray * 1)()
    assert not lib.anofox_ridge_fit(y, xs, 1, ropt, C.byref(core), None, C.byref(err))
    assert err.code == abi.ERROR_INVALID_ALPHA


def test_host_prediction_helpers_need_no_gpu():
    """anofox_t_critical / anofox_predict_with_interval are scalar host helpers (like aic / bic)."""
    from scipy import stats as sps
    import oracle
    pkg = import_pkg()
    for df in (1, 2, 7, 30, 400):
        assert abs(pkg.t_critical(0.95, df) / sps.t.ppf(0.975, df) - 1.0) < 1e-10
    got = pkg.predict_with_interval([2.0, float("nan")], 1.0, [3.0, 5.0], 0.5, 20, 0.95)
    ok, want = oracle.predict_with_interval([2.0, float("nan")], 1.0, [3.0, 5.0], 0.5, 20, 0.95)
    assert ok and abs(got["yhat_lower"] - want[1]) < 1e-12 and abs(got["yhat_upper"] - want[2]) < 1e-12
    abi = import_pkg("_abi")
    assert not abi.load().anofox_predict_with_interval(None, 0, 0.0, None, 0, 1.0, 10, 0.95,
                                                        C.byref(abi.AnofoxPredictionResult()))


def test_missing_rccl_is_a_clean_error_not_a_crash():
    """The library loads RCCL at run time (dlopen).  When it cannot — a host without RCCL — the first comm call must
    fail with ANOFOX_ERROR_INTERNAL and a message, not crash (round 2's error path read dlerror() twice: the second
    read is NULL).  Run in a child process: the loader result is cached per process."""
    import subprocess
    import sys
    code = (
        "import importlib, ctypes as C\n"
        "abi = importlib.import_module('anofox-statistics_amd._abi')\n"
        "lib = abi.load()\n"
        "err = abi.AnofoxError()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "ok = lib.anofox_hip_comm_unique_id(buf, C.byref(err))\n"
        "assert not ok and err.code == 99, (ok, err.code)\n"
        "assert 'RCCL is not available' in err.text(), err.text()\n"
        "ok = lib.anofox_hip_comm_unique_id(buf, C.byref(err))   # and again: the cached failure\n"
        "assert not ok and 'RCCL is not available' in err.text()\n"
        "assert lib.anofox_hip_comm_ranks_seen(None) == 0\n"
        "print('clean')\n")
    env = dict(os.environ, ANOFOX_RCCL_LIB="/nonexistent/librccl.so.1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "clean" in out.stdout, out.stderr[-2000:]
