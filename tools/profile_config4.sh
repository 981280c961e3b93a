#!/bin/bash
# The configs[4] shard under rocprofv3 (on the GPU box, via gpurun): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their
# own passes (MI355X_MICROARCH.md, HBM section).  usage: tools/profile_config4.sh <tag> [key=value ... of tools/profile_config4.py]
set -o pipefail
TAG=${1:-r05_config4}; shift
EXTRA="$*"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
run() {  # name, reps, rocprofv3 args...
  local name=$1 reps=$2; shift 2
  rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $REPO/tools/profile_config4.py $reps $EXTRA > $OUT/run_$name.json 2> $OUT/run_$name.err
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $OUT/run_$name.err; exit $rc; fi
}
run stats 5 --stats
run pmc_fetch 1 --pmc FETCH_SIZE
run pmc_write 1 --pmc WRITE_SIZE
cd $REPO && python3 tools/summarize_config4.py $TAG
