"""CPU tier: AddressSanitizer + UndefinedBehaviorSanitizer builds of the oracle's C code and of the library's host
code (GPU ASan is not available on this pool).  tests/tools/Makefile compiles csrc/host_api.hip and
csrc/agg_state.hip as plain C++ with g++ (kernel launchers stubbed: nothing reaches them without a GPU) and
oracle/anofox_oracle.c with gcc, each with a driver; the DuckDB shim's arena (duckdb_shim/agg_arena.hpp) runs against a
recording mock of the C ABI (tests/tools/arena_sanitize.cpp: Update vectors of three threads, flushes, Combine, Finalize,
narrow and wide designs, concurrent writers, the windowed-aggregate protocol, a failing device; every accepted row must
reach the right slot in the reference's order), and the DuckDB glue itself (duckdb_shim/fit_agg_hip.cpp) is compiled
against a stand-in of DuckDB's headers (tests/tools/duckdb_stub) with -Wall -Wextra -Werror and driven through
registration, bind and the Update / Combine / Finalize / Destroy protocol as a parallel hash aggregate, as the naive
window aggregator (the reference's test/sql/comprehensive_tests.test:425-444) and as a segment tree
(tests/tools/glue_sanitize.cpp).  A non-zero exit or any sanitizer report fails the test."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

OUT = "/tmp/anofox_sanitize"


@pytest.fixture(scope="module")
def built():
    if not shutil.which("g++") or not os.path.isdir("/opt/rocm/include/hip"):
        pytest.skip("needs g++ and the HIP headers")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "tools"), f"OUT={OUT}"], stdout=subprocess.DEVNULL)
    return OUT


@pytest.mark.parametrize("unit", ["oracle_sanitize", "host_sanitize", "arena_sanitize", "glue_sanitize"])
def test_sanitizer_unit(built, unit):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(built, unit)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    assert "all" in r.stdout


@pytest.mark.parametrize("unit", ["arena_tsan", "glue_tsan"])
def test_thread_sanitizer_unit(built, unit):
    """The DuckDB shim from several threads under ThreadSanitizer (mock C ABI): no data race in the arena's per-thread
    chunks / shipping lock / slot hand-out, nor in the glue's per-width arena set."""
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(built, unit)], env=env, capture_output=True, text=True, timeout=600)
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this environment")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr
    assert "all" in r.stdout
