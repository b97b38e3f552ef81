#!/usr/bin/env bash
# Round-4 profiles of cfg5 (50 000 x 4096 x 128, inference): kernel stats of bench.py and of one native slab, FETCH / WRITE of the
# solve_tiles kernel (its scratch traffic).  Run on the gpurun box from the repo root.
set -u
OUT=$PWD/gpurun_out/prof_r04c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NB=$R/anofox-statistics_amd/csrc/tools/native_bench
echo "== kernel stats: bench cfg5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_cfg5 -o ks -- python3 $R/bench.py --groups 50000 --rows 4096 --features 128 --inference --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_under_rocprof.json 2> $OUT/ks_cfg5.err
echo "== kernel stats: native cfg5 slab"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_slab -o ks -- $NB 13786 4096 128 ols 3 inference > $OUT/native_slab.json 2> $OUT/ks_slab.err
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  name=${pass%% *}; ctrs=${pass#* }
  echo "== pmc $name"
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- $NB 13786 4096 128 ols 2 inference > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
find $OUT -name "*stats.csv" | head
