#!/usr/bin/env bash
# A/B of the fine LDS-DMA ring (ANOFOX_QUAD_FINE=<ring depth>) against the default dispatch, one box: bench.py prints the accumulate kernel's
# time from the library's HIP events and the parity of the fits against the oracle.
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for p in ${WIDTHS:-33 40 48 50 56 64}; do
  g=50000; [ $p -le 42 ] && g=100000
  python bench.py --groups $g --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/qfine.err | w "p=$p default"
  for ring in ${RINGS:-4 3}; do
    ANOFOX_QUAD_FINE=$ring python bench.py --groups $g --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/qfine.err | w "p=$p fine$ring"
  done
done
