#!/usr/bin/env bash
# what-if builds of accumulate_wide's speculative steady state (csrc/accumulate_wide_impl.h ANOFOX_WIDE_SKIP; results are wrong on purpose):
# which part sets the kernel's time at 3-4 column tiles?  accumulate_ms of tools/native_bench per variant.
B=$PWD/anofox-statistics_amd
for a in "50000 1000 48" "50000 1000 64"; do
  echo "== $a default"; $B/csrc/tools/native_bench $a ols 5
  for v in ${VARIANTS:-1 2 4 3 6 7}; do
    echo "== $a skip=$v"; LD_LIBRARY_PATH=$B/whatif/v$v:$LD_LIBRARY_PATH $B/csrc/tools/native_bench $a ols 5
  done
done
