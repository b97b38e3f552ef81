#!/usr/bin/env bash
# same-box A/B: accumulate_wide at 3-4 column tiles with EIGHT wavefronts per workgroup (-DANOFOX_WIDE_WAVES=8 for accumulate_wide.hip) against four
L=$PWD/anofox-statistics_amd
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for p in ${PS:-44 48 56 64}; do
  for v in default w8 default w8; do
    lib=$L/whatif/$v/libanofox_stats_hip.so; [ $v = default ] && lib=$L/libanofox_stats_hip.so
    ANOFOX_STATS_HIP_LIB=$lib python bench.py --groups 47000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/w8.err | w "p=$p $v"
  done
done
