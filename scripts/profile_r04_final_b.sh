#!/usr/bin/env bash
# Round-4 end-of-round measurements, part B: the widths at n = 1000 with the speculative kernels on / off ON ONE BOX, cfg2 / cfg3,
# the arena and the ingest path, PMC of the LDS-DMA kernel.
set -u
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
w() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'step', round(d['ms_per_step'],3), 'ms  kernel', r['kernel'], round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s', 'frac_hbm', round(r.get('kernel_frac_of_hbm_peak', r['kernel_frac']),3), 'parity', d['parity']['ok'])" "$1"; }
for p in 9 12 16 20 24 26 27 28 30 32 33 34 36 38 40 42; do
  G=100000
  python bench.py --groups $G --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>$OUT/w.err | w "p=$p spec=on " | tee -a $OUT/widths.txt
  ANOFOX_QUAD_SPEC=0 python bench.py --groups $G --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>$OUT/w.err | w "p=$p spec=off" | tee -a $OUT/widths.txt
done
for p in 48 64 96 128; do
  python bench.py --groups 50000 --features $p --no-cpu-baseline --steps 10 --warmup 3 2>>$OUT/w.err | w "p=$p (50k groups)" | tee -a $OUT/widths.txt
done
echo "== cfg2 / cfg3"
python bench.py --groups 10000 --no-cpu-baseline --no-end-to-end --steps 200 --warmup 20 > $OUT/bench_cfg2.json 2>>$OUT/w.err; w cfg2 < $OUT/bench_cfg2.json | tee -a $OUT/widths.txt
python bench.py --model ridge --no-cpu-baseline --no-end-to-end > $OUT/bench_cfg3_ridge.json 2>>$OUT/w.err; w cfg3_ridge < $OUT/bench_cfg3_ridge.json | tee -a $OUT/widths.txt
python bench.py --model wls --no-cpu-baseline --no-end-to-end > $OUT/bench_cfg3_wls.json 2>>$OUT/w.err; w cfg3_wls < $OUT/bench_cfg3_wls.json | tee -a $OUT/widths.txt
python bench.py --inference --no-cpu-baseline > $OUT/bench_inference.json 2>>$OUT/w.err; w ols_inference < $OUT/bench_inference.json | tee -a $OUT/widths.txt
echo "== arena (row log on: the default) and ingest"
for t in 4 8 16 32; do anofox-statistics_amd/duckdb_shim/arena_bench $t 2>&1 | tail -1 | tee -a $OUT/arena.jsonl; done
for t in 8 16 32; do ANOFOX_HIP_RETAIN_BYTES=0 ANOFOX_HIP_RETAIN_HOST_BYTES=0 anofox-statistics_amd/duckdb_shim/arena_bench $t 2>&1 | tail -1 | sed 's/^{/{"row_log": "off", /' | tee -a $OUT/arena.jsonl; done
python scripts/ingest_bench.py --groups 1000000 --rows 64 --steps 2 2>>$OUT/w.err | tee $OUT/ingest.jsonl | cut -c1-400
cd /tmp && export TMPDIR=/tmp
NB=$R/anofox-statistics_amd/csrc/tools/native_bench
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64" "insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  name=${pass%% *}; ctrs=${pass#* }
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc33_$name -o pmc -- $NB 100000 1000 33 ols 2 > $OUT/pmc33_$name.json 2> $OUT/pmc33_$name.err || echo "pass $name failed"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks33 -o ks -- $NB 100000 1000 33 ols 3 > $OUT/ks33.json 2> $OUT/ks33.err
ls $OUT | head -40
