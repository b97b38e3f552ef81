#!/usr/bin/env bash
# Round-3 profiles (run on the gpurun box from the repo root; outputs under gpurun_out/prof_r03, summaries copied into profiles/ afterwards):
#   kernel stats (rocprofv3 --kernel-trace --stats) of the two bench lines, PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE, MFMA /
#   instruction counters) of the torch-free native_bench on the cfg5 slab shape, where the new solve_tiles kernel runs.
set -u
OUT=$PWD/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NB=$R/anofox-statistics_amd/csrc/tools/native_bench
echo "== kernel stats: bench default"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_default -o ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_default_under_rocprof.json 2> $OUT/ks_default.err
echo "== kernel stats: cfg5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_cfg5 -o ks -- python3 $R/bench.py --groups 50000 --rows 4096 --features 128 --inference --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_under_rocprof.json 2> $OUT/ks_cfg5.err
echo "== kernel stats: native cfg5 slab"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_slab -o ks -- $NB 13786 4096 128 ols 3 inference > $OUT/native_slab.json 2> $OUT/ks_slab.err
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VMEM_RD"; do
  name=${pass%% *}; ctrs=${pass#* }
  echo "== pmc $name: $ctrs"
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- $NB 13786 4096 128 ols 2 inference > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
echo "== pmc narrow fetch / write (metric config)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_narrow_fetch -o pmc -- $NB 1000000 1000 8 ols 2 > $OUT/pmc_narrow_fetch.json 2> $OUT/pmc_narrow_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_narrow_write -o pmc -- $NB 1000000 1000 8 ols 2 > $OUT/pmc_narrow_write.json 2> $OUT/pmc_narrow_write.err
find $OUT -name "*.csv" | head -40
