#!/usr/bin/env bash
# same-box A/B of accumulate_wide's speculative staging: 16 rows x 8 columns per load instruction (128-byte runs per column, default)
# against 32 rows x 4 columns (256-byte runs; csrc: make variant VARNAME=run256 VARFLAGS="'-DANOFOX_WIDE_RUN256(T)=1'")
L=$PWD/anofox-statistics_amd
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for p in ${PS:-44 48 56 64 96 128}; do
  for v in default run256 default run256; do
    lib=$L/libanofox_stats_hip_$v.so; [ $v = default ] && lib=$L/libanofox_stats_hip.so
    ANOFOX_STATS_HIP_LIB=$lib python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wrun.err | w "p=$p $v"
  done
done
