#!/usr/bin/env bash
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for p in 43 48 56 64; do
  for v in 0 32 64; do
    ANOFOX_WIDE_DMA=$v python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wdma64.err | w "p=$p dma=$v"
  done
done
