#!/usr/bin/env bash
# Run GPU stages one after another on the gpurun box.  Each stage has its own timeout; an ordinary failure lets
# the next stage run (its log is kept), a timeout/kill (124/137) stops everything.
# usage: scripts/gpu_stages.sh "name|timeout_s|command" ...
set -u
mkdir -p gpurun_out
cd /tmp 2>/dev/null && export TMPDIR=/tmp; cd - >/dev/null
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== stage $name (timeout ${tmo}s): $cmd" | tee -a gpurun_out/stages.log
  t0=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/${name}.log" 2>&1
  rc=$?
  echo "=== stage $name rc=$rc in $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/stages.log
  tail -n 15 "gpurun_out/${name}.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "=== stage $name timed out / was killed: stopping" | tee -a gpurun_out/stages.log
    exit $rc
  fi
done
exit 0
