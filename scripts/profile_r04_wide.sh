#!/usr/bin/env bash
# Round-4 PMC passes of accumulate_wide_kernel<3> / <4> (33 <= p <= 64 at n = 1000) — VERDICT r3 item 3b.  Run on the gpurun box from the
# repo root; outputs under gpurun_out/prof_r04w (summaries are copied into profiles/ afterwards).  Counters in their own passes, with
# --kernel-trace only (never --stats / sys-trace together with --pmc).
set -u
OUT=$PWD/gpurun_out/prof_r04w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NB=$R/anofox-statistics_amd/csrc/tools/native_bench
for p in 33 48 64; do
  echo "== kernel stats p=$p"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_p$p -o ks -- $NB 100000 1000 $p ols 3 > $OUT/ks_p$p.json 2> $OUT/ks_p$p.err
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64" "insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    name=${pass%% *}; ctrs=${pass#* }
    echo "== pmc p=$p $name: $ctrs"
    rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_p${p}_$name -o pmc -- $NB 100000 1000 $p ols 2 > $OUT/pmc_p${p}_$name.json 2> $OUT/pmc_p${p}_$name.err || echo "   (pass failed: see pmc_p${p}_$name.err)"
  done
done
find $OUT -name "*.csv" | head -60
