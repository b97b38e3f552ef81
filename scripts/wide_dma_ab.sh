#!/usr/bin/env bash
# A/B of accumulate_wide's speculative kernel with LDS-DMA staging (ANOFOX_WIDE_DMA=1) on one box
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', r['kernel'], round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for p in 43 48 49 56 64; do
  python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wdma.err | w "p=$p dma=0"
  ANOFOX_WIDE_DMA=1 python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wdma.err | w "p=$p dma=1"
done
for p in 36 40 42; do
  python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wdma.err | w "p=$p quad-dma"
  ANOFOX_QUAD_SPEC_MAXP=34 ANOFOX_WIDE_DMA=1 python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wdma.err | w "p=$p wide-dma"
done
