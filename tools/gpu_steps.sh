#!/bin/bash
# Helper for one gpurun call made of several steps: every step runs under its own `timeout -k 10`, writes
# gpurun_out/<name>.log, and a step that was KILLED at its limit ends the call (no further GPU step after a hang).
#   source tools/gpu_steps.sh; step NAME SECONDS cmd...
mkdir -p gpurun_out
step() {
  local name=$1 limit=$2
  shift 2
  echo "== $name: $*"
  timeout -k 10 "$limit" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "   rc=$rc"
  tail -n 4 "gpurun_out/$name.log" | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "   $name hit its time limit: no further GPU step in this call"
    exit 1
  fi
  return 0
}
