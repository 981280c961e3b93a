#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command, then HBM traffic counters
# in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py "$@" --skip-gn-full-loop > $OUT/bench_stats.json 2> $OUT/bench_stats.err
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --skip-single-row --skip-gn-full-loop > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --skip-single-row --skip-gn-full-loop > $OUT/bench_write.json 2> $OUT/bench_write.err
echo "write rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --skip-single-row --skip-gn-full-loop > $OUT/bench_sq.json 2> $OUT/bench_sq.err
echo "sq rc=$?"
find $OUT -name "*.csv" | head -30
