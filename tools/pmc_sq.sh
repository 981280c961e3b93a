#!/bin/bash
# SQ counter pass for one short bench run (own pass: no other tracing domains).
TAG=${1:-sq}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-single-row --skip-quadrature --iters 10 "$@" > $OUT/a.json 2> $OUT/a.err
rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-single-row --skip-quadrature --iters 10 "$@" > $OUT/b.json 2> $OUT/b.err
rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
python3 - <<PY
import csv, glob, collections
for sub in ('a','b'):
    for f in glob.glob('$OUT/%s/*/*_counter_collection.csv' % sub):
        agg = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'rows4' in r['Kernel_Name'] or ('gn_' in r['Kernel_Name'] and 'tables' not in r['Kernel_Name']):
                agg[(r['Kernel_Name'][:28], r['Counter_Name'])] += float(r['Counter_Value'])
        for k in sorted(agg): print(k, '%.4g' % agg[k])
PY
