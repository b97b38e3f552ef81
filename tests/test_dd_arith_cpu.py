"""CPU tier: the error-free transformations of the refinement passes (csrc/dd_arith.h).

hipcc's default -ffp-contract=fast-honor-pragmas may fuse a product with the sum that consumes it across statements;
inside two_sum that breaks the compensation and the "double-double" residual carries working-precision noise (round 3:
a 15 x 15 system of the deep fuzz sweep wandered between 4e-10 and 1.4e-8).  Two checks that need no GPU:
  * the formulas themselves, compiled for the host, are exact against __float128 (tests/tools/dd_arith_check.hip);
  * the DEVICE code hipcc generates for gfx950 keeps one v_fma_f64 per term (two_prod's) — and the same formulas without
    the guard do not, i.e. the probe sees the defect the guard removes (tests/tools/dd_arith_probe.hip)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

OUT = "/tmp/anofox_sanitize"
TOOLS = os.path.join(ROOT, "tests", "tools")
CSRC = os.path.join(ROOT, "anofox-statistics_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _need_hipcc():
    if not os.path.exists(HIPCC):
        pytest.skip("needs hipcc")
    os.makedirs(OUT, exist_ok=True)


def test_dd_arith_host():
    _need_hipcc()
    exe = os.path.join(OUT, "dd_arith_check")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fno-fast-math", "-mfma", "--offload-arch=gfx950", "-I" + CSRC,
                           os.path.join(TOOLS, "dd_arith_check.hip"), "-o", exe], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def _fma_count(extra):
    asm = os.path.join(OUT, "dd_probe%s.s" % ("_unguarded" if extra else ""))
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fno-fast-math", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                           "-I" + CSRC, *extra, os.path.join(TOOLS, "dd_arith_probe.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    text = open(asm).read()
    return text.count("v_fma_f64"), text.count("v_mul_f64")


def test_dd_arith_device_code():
    _need_hipcc()
    fma, mul = _fma_count([])
    assert (fma, mul) == (1, 1), f"two_sum / two_prod of dd_arith.h: expected one v_fma_f64 and one v_mul_f64 per term, got {fma} / {mul}"
    fma_unguarded, _ = _fma_count(["-DDD_UNGUARDED"])
    if fma_unguarded <= fma:
        pytest.skip("this hipcc does not contract the unguarded formulas: the probe cannot show the defect")
    assert fma_unguarded > fma
