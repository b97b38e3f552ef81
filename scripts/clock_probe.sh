#!/usr/bin/env bash
# shader clock and package power while a width runs (rocm-smi polled in the background): is the clock what differs between widths / boxes?
for spec in "8 1000000" "16 100000" "33 100000" "48 50000" "64 50000" "128 12000"; do
  set -- $spec; p=$1; G=$2
  ( for i in $(seq 1 14); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|Average Graphics Package Power|Current Socket" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clk_$p.txt &
  MON=$!
  python bench.py --groups $G --features $p --no-cpu-baseline --no-end-to-end --steps 400 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('p=$p', round(r['kernel_ms_per_step'],3), 'ms', round(r['hbm_GBps_algorithmic']), 'GB/s')"
  wait $MON
  echo "--- p=$p samples:"; sort gpurun_out/clk_$p.txt | uniq -c | sort -rn | head -4
done
