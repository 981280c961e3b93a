#!/bin/bash
# FETCH_SIZE (L2 -> fabric read requests) of the traversal kernel variants, one rocprofv3 --pmc pass (own pass, no
# other tracing domains).  usage: tools/pmc_tiled.sh <tag> [bench_tiled.py args...]
TAG=${1:-tiled}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $REPO/tools/bench_tiled.py --reps 1 "$@" > $OUT/run.txt 2> $OUT/run.err
rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/fetch/*/*_counter_collection.csv'):
    rows = [r for r in csv.DictReader(open(f)) if 'rows' in r['Kernel_Name']]
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace('void dexct::', '')
        print('%-40s dispatch %6s  FETCH_SIZE %.2f GB (x2 for dword-per-lane loads: %.2f GB)' % (
            name, r.get('Dispatch_Id', '?'), float(r['Counter_Value']) * 1024 / 1e9, 2 * float(r['Counter_Value']) * 1024 / 1e9))
PY
cat $OUT/run.txt
