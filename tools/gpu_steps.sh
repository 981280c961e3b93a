#!/bin/bash
# gpu_steps.sh <log-prefix> <timeout-seconds> '<cmd 1>' '<cmd 2>' ...: the commands one after the other on the GPU box, each under
# its own `timeout -k 10`, stdout + stderr of step k to gpurun_out/<prefix>_<k>.log.  An ordinary failure (an assertion) does not
# stop the sequence; a step that was KILLED at its limit does (no further GPU step after a hang).
prefix=$1; limit=$2; shift 2
mkdir -p gpurun_out
k=0; worst=0
for cmd in "$@"; do
  k=$((k + 1))
  echo "== step $k: $cmd" | tee gpurun_out/${prefix}_$k.log
  timeout -k 10 "$limit" bash -c "$cmd" >> gpurun_out/${prefix}_$k.log 2>&1
  rc=$?
  echo "== step $k exit code $rc" | tee -a gpurun_out/${prefix}_$k.log
  tail -n 3 gpurun_out/${prefix}_$k.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $k hit its limit: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && worst=$rc
done
exit $worst
