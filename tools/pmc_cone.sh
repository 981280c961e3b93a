#!/bin/bash
# SQ counters of the cone kernels (two rocprofv3 --pmc passes, nothing else traced).  usage: tools/pmc_cone.sh <tag> [views]
TAG=${1:-cone}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/tools/bench_cone.py "$@" > $OUT/a.txt 2> $OUT/a.err
rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/tools/bench_cone.py "$@" > $OUT/b.txt 2> $OUT/b.err
rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
python3 - <<PY
import csv, glob, collections
for sub in ('a', 'b'):
    for f in glob.glob('$OUT/%s/*/*_counter_collection.csv' % sub):
        last = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if 'cone_' not in r['Kernel_Name'] or 'layout' in r['Kernel_Name']:
                continue
            name = r['Kernel_Name'].split('(')[0].replace('void dexct::', '')
            last.setdefault(name, {})[r['Counter_Name']] = float(r['Counter_Value'])
        for name, c in last.items():
            print(name, ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
PY
grep "cone kernel" $OUT/a.txt
