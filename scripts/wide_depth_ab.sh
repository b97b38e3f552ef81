#!/usr/bin/env bash
# same-box A/B of accumulate_wide builds (csrc/Makefile `variant`): tile deal and staging depth
L=$PWD/anofox-statistics_amd
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'])" "$1"; }
for p in ${PS:-43 48 56 64}; do
  for v in nosplit_d1 d1 d2 default d4; do
    lib=$L/libanofox_stats_hip_$v.so; [ $v = default ] && lib=$L/libanofox_stats_hip.so
    ANOFOX_STATS_HIP_LIB=$lib python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wdepth.err | w "p=$p $v"
  done
done
