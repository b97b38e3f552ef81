"""Run single cases of the fuzz families by seed (GPU box): python scripts/run_fuzz_seeds.py <narrow|wide|very> <seed> [<seed> ...]"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_fuzz as F
from conftest import import_pkg
pkg = import_pkg(); ctx = pkg.Context()
kind = {"narrow": False, "wide": True, "very": "very"}[sys.argv[1]]
bad = 0
for a in sys.argv[2:]:
    try:
        F._run(pkg, ctx, int(a), kind)
        print("seed", a, "ok")
    except AssertionError as e:
        bad += 1
        print("seed", a, "FAILED:", str(e).splitlines()[0])
sys.exit(1 if bad else 0)
