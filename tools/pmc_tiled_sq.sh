#!/bin/bash
# SQ counters of the traversal kernel variants (two rocprofv3 --pmc passes, nothing else traced).
# usage: tools/pmc_tiled_sq.sh <tag> [bench_tiled.py args...]
TAG=${1:-tiledsq}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/tools/bench_tiled.py --reps 1 "$@" > $OUT/a.txt 2> $OUT/a.err
rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/tools/bench_tiled.py --reps 1 "$@" > $OUT/b.txt 2> $OUT/b.err
rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
python3 - <<PY
import csv, glob, collections
for sub in ('a', 'b'):
    for f in glob.glob('$OUT/%s/*/*_counter_collection.csv' % sub):
        seen = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if 'rows' not in r['Kernel_Name']:
                continue
            name = r['Kernel_Name'].split('(')[0].replace('void dexct::', '')
            key = (name, r['Dispatch_Id'])
            seen.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
        last = {}
        for (name, d), c in seen.items():
            last[(name, tuple(sorted(c)))] = (d, c)          # keep the last dispatch of each kernel name
        for (name, _), (d, c) in last.items():
            print(name, 'dispatch', d, ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
PY
cat $OUT/a.txt
