#!/usr/bin/env bash
# kernel stats + HBM traffic of accumulate_quad (p = 16 and 24, 100 000 groups x 1000 rows); outputs under gpurun_out/prof_quad
set -u
export OUT=$PWD/gpurun_out/prof_quad
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NB=$R/anofox-statistics_amd/csrc/tools/native_bench
for p in 16 24; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_p$p -o ks -- $NB 100000 1000 $p ols 5 > $OUT/native_p$p.json 2> $OUT/ks_p$p.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_p$p -o pmc -- $NB 100000 1000 $p ols 2 > /dev/null 2> $OUT/fetch_p$p.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_p$p -o pmc -- $NB 100000 1000 $p ols 2 > /dev/null 2> $OUT/write_p$p.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d $OUT/mfma_p$p -o pmc -- $NB 100000 1000 $p ols 2 > /dev/null 2> $OUT/mfma_p$p.err
done
ls $OUT
