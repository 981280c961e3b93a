#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command, then HBM traffic counters
# in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with the
# hip / hsa trace domains).  A pass that fails stops the script with its exit code (no summary from partial CSVs).
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
# the PMC passes keep the single-row and cone-beam legs of bench.py (their kernels get rooflines too) and drop the
# CPU baseline, the public-boundary timing and the extra Newton launches
# (--skip-noisy: the noisy step runs the traversal kernel with other outputs - path lengths for the per-bin Poisson sampler - and
# would be averaged into the step kernel's counters; its own passes: tools/profile_config4.sh <tag> noisy=1 ...)
LEAN="--steps 1 --warmup 0 --no-cpu-baseline --skip-gn-full-loop --skip-dropin --skip-quadrature --skip-noisy"
run() {  # name, rocprofv3 args...
  local name=$1; shift
  rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $REPO/bench.py $BENCH_ARGS > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $OUT/bench_$name.err; exit $rc; fi
}
BENCH_ARGS="$* --skip-gn-full-loop --skip-dropin --skip-quadrature"
run stats --stats
BENCH_ARGS="$* $LEAN"
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES
find $OUT -name "*.csv" | head -30
