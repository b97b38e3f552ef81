#!/usr/bin/env bash
# same-box A/B of accumulate_wide's speculative version at 3-4 column tiles: register budget for 3 wavefronts per SIMD
# (-DANOFOX_WIDE_FAST_WPS(T)=3: T = 3 -> 124 registers / occupancy 4, T = 4 -> 168 / 3) against the default (162 / 3, 216 / 2)
L=$PWD/anofox-statistics_amd
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for p in ${PS:-44 48 56 64}; do
  for v in default w3 w3d1 default w3; do
    lib=$L/whatif/$v/libanofox_stats_hip.so; [ $v = default ] && lib=$L/libanofox_stats_hip.so
    ANOFOX_STATS_HIP_LIB=$lib python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/wocc.err | w "p=$p $v"
  done
done
