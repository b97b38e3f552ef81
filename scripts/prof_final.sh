#!/usr/bin/env bash
# kernel stats (rocprofv3 --kernel-trace --stats) of the bench lines at the end of round 3: default, cfg5, window expanding
set -u
OUT=$PWD/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== default"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_default -o ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_default_under_rocprof.json 2> $OUT/ks_default.err
echo "== cfg5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_cfg5 -o ks -- python3 $R/bench.py --groups 50000 --rows 4096 --features 128 --inference --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_under_rocprof.json 2> $OUT/ks_cfg5.err
echo "== window expanding"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_wexp -o ks -- python3 $R/bench.py --window --groups 1000000 --rows 100 --features 3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_wexp_under_rocprof.json 2> $OUT/ks_wexp.err
find $OUT -name "*kernel_stats.csv"
