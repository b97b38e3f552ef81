"""Diagnostic: where a wavefront of accumulate_wide_kernel spends its chunk loop, from the stamp build
(make -C anofox-statistics_amd/csrc diag).  usage: python scripts/dbg_acc_stamps.py [p] [groups] [rows]"""
import os, sys, ctypes as C, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ANOFOX_STATS_HIP_LIB"] = os.path.join(ROOT, "anofox-statistics_amd", "libanofox_stats_hip_diag.so")
import torch
pkg = importlib.import_module("anofox-statistics_amd")
synth = importlib.import_module("anofox-statistics_amd.synth")
abi = importlib.import_module("anofox-statistics_amd._abi")
p = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
offs, y, xc, _ = synth.make_grouped(G, n, p, device="cuda", chunk_groups=64)
ctx = pkg.Context()
opts = pkg.RegressionOptions().batch_options("ols")
for _ in range(3):
    ctx.fit_batch_device(offs, y, xc, None, opts)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib = abi.load()
if p <= 64:  # the kernels of 1 .. 4 column tiles live in another translation unit; they also leave milestones
    lib.anofox_hip_diag_acc_stamps_t4.restype = C.c_int
    print("rc", lib.anofox_hip_diag_acc_stamps_t4(buf))
else:
    lib.anofox_hip_diag_acc_stamps.restype = C.c_int
    print("rc", lib.anofox_hip_diag_acc_stamps(buf))
s = list(buf)
nch = max(s[6], 1)
names = ["row masks of the chunk read from LDS", "next chunk's global loads issued (+ rare repairs)",
         "slabs: fragment reads + MFMAs + side sums", "s_waitcnt vmcnt(0): next chunk's loads landed",
         "next chunk staged into LDS (filter, shift, ds_write, ballots)", "wait at the barrier"]
print(f"p={p} G={G} n={n}: {nch} chunks, {s[7]} ticks for the group = {s[7]/nch:.0f} per chunk")
for k in range(6):
    print(f"  {names[k]:<64} {s[k]/nch:8.0f} ticks/chunk  {100.0*s[k]/max(s[7],1):5.1f} %")
if p <= 64:
    marks = ["setup + first barrier", "first row known (the shift)", "chunk 0 staged, barrier", "steady trips done", "chunk loop done",
             "split tiles collected", "speculation checked", "record written"]
    prev = 0
    print("milestones of the workgroup in the middle of the grid (ticks since entry; 100 ticks = 1 us):")
    for k in range(8):
        print(f"  {marks[k]:<40} {s[8 + k]:8d}  (+{s[8 + k] - prev})")
        prev = s[8 + k]
