#!/usr/bin/env bash
# A/B of the LDS-DMA accumulate_quad ring depth (ANOFOX_QUAD_RING=3) on one box
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', r['kernel'], round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'])" "$1"; }
for p in ${RING_PS:-27 30 31 33 34 35 36 38 40 42}; do
  python bench.py --groups 100000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/ring.err | w "p=$p ring=2"
  ANOFOX_QUAD_RING=${RING_B:-3} python bench.py --groups 100000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/ring.err | w "p=$p ring=${RING_B:-3}"
done
