#!/usr/bin/env bash
# One gpurun call = the stage list in scripts/gpu_stages.txt (one "name|timeout|command" per line).
set -u
mapfile -t STAGES < <(grep -v '^\s*#' scripts/gpu_stages.txt | grep -v '^\s*$')
exec scripts/gpu_stages.sh "${STAGES[@]}"
