#!/usr/bin/env bash
# same-box A/B of the slab sizes of the wide path (host_api.hip: a remainder below a quarter of a slab takes groups from the slab before it)
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  fits/s', round(d['value']), ' kernel', round(r['kernel_ms_per_step'],3), 'ms', 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'])" "$1"; }
for cfg in "50000 64" "50000 56" "50000 50" "100000 64" "80000 48"; do
  set -- $cfg
  for v in 0 1 0 1; do
    ANOFOX_WIDE_SLAB_BALANCE=$v python bench.py --groups $1 --features $2 --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/slabab.err | w "G=$1 p=$2 balance=$v"
  done
done
