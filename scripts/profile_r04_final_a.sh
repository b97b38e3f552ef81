#!/usr/bin/env bash
# Round-4 end-of-round measurements, part A (run on the gpurun box from the repo root; summaries are copied into profiles/).
set -u
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[2], 'fits/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median'],3), 'kernel', r['kernel'], 'ms', round(r['kernel_ms_per_step'],3), 'min', round(r['kernel_ms_min'],3), 'bound', r['bound'], 'frac', round(r['frac'],4), 'kernel_frac', round(r['kernel_frac'],4), 'GB/s', round(r['hbm_GBps_algorithmic']), 'parity', d['parity']['ok'], d['parity']['max_coef_rel_err'], d['parity']['max_diag_rel_err'])" "$1" "$2"; }
echo "== bench default (with end_to_end, row-log variant and cpu_baseline)"
python bench.py --end-to-end-row-log > $OUT/bench_default.json 2> $OUT/bench_default.err && line $OUT/bench_default.json default
python3 -c "import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print('end_to_end', {k: d['end_to_end'].get(k) for k in ('fits_per_s','rows_per_s','GBps_pcie','seconds','parity')}); print('end_to_end_with_row_log', {k: d['end_to_end_with_row_log'].get(k) for k in ('fits_per_s','rows_per_s','GBps_pcie','seconds','parity')}); print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
echo "== cfg5 A/B on this box"
for v in "default" "ANOFOX_SOLVE_PARK=0" "ANOFOX_WIDE_SPLIT=1" "ANOFOX_WIDE_OVERLAP=0"; do
  if [ "$v" = default ]; then e=""; else e="$v"; fi
  env $e python bench.py --groups 50000 --rows 4096 --features 128 --inference --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_cfg5_$v.json 2> $OUT/bench_cfg5_$v.err && line $OUT/bench_cfg5_$v.json "cfg5[$v]"
done
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats: bench default"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_default -o ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $OUT/bench_default_under_rocprof.json 2> $OUT/ks_default.err
echo "== kernel stats: bench cfg5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_cfg5 -o ks -- python3 $R/bench.py --groups 50000 --rows 4096 --features 128 --inference --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_under_rocprof.json 2> $OUT/ks_cfg5.err
head -6 $OUT/ks_default/ks_kernel_stats.csv | cut -c1-180
