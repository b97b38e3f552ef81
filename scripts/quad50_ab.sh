#!/usr/bin/env bash
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', r['kernel'], round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for p in 43 44 46 48 49 50; do
  ANOFOX_QUAD_SPEC_MAXP=42 python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/q50.err | w "p=$p wide"
  ANOFOX_QUAD_SPEC_MAXP=50 python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/q50.err | w "p=$p quad"
done
