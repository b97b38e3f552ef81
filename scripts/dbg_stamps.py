"""Diagnostic: phase breakdown of solve_wide_kernel from the stamp build (make -C anofox-statistics_amd/csrc diag)."""
import os, sys, ctypes as C, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ANOFOX_STATS_HIP_LIB"] = os.path.join(ROOT, "anofox-statistics_amd", "libanofox_stats_hip_diag.so")
import torch
pkg = importlib.import_module("anofox-statistics_amd")
synth = importlib.import_module("anofox-statistics_amd.synth")
abi = importlib.import_module("anofox-statistics_amd._abi")
G, n, p = 512, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 128
inference = len(sys.argv) > 2
offs, y, xc, _ = synth.make_grouped(G, n, p, device="cuda", chunk_groups=64)
ctx = pkg.Context()
opts = pkg.RegressionOptions(compute_inference=inference).batch_options("ols")
for _ in range(2):
    ctx.fit_batch_device(offs, y, xc, None, opts)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib = abi.load()
lib.anofox_hip_diag_solve_stamps.restype = C.c_int
print("rc", lib.anofox_hip_diag_solve_stamps(buf))
s = list(buf)
names = {0: "start", 1: "loaded", 2: "cholesky done", 3: "inverse done", 4: "beta/diag done", 5: "core written", 6: "end"}
for k in range(1, 7):
    print(f"{names[k]:>16}: +{s[k]-s[k-1]:8d} ticks")
print("chol block 0: diag", s[9]-s[8], "panel", s[10]-s[9], "update", s[11]-s[10])
print("total", s[6]-s[0], "ticks")
