#!/bin/bash
# L1 (TCP) counters of the traversal kernel variants, one rocprofv3 --pmc pass per group.
# usage: tools/pmc_tiled_mem.sh <tag> [bench_tiled.py args...]
TAG=${1:-tiledmem}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           ; do
  # (a third group - TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum -
  # made rocprofv3 abort and the run hang on this pool in round 2: do not add TA counters back)
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/tools/bench_tiled.py --reps 1 "$@" > $OUT/g$i.txt 2> $OUT/g$i.err
  echo "group $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
last = collections.OrderedDict()
for f in sorted(glob.glob('$OUT/g*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'rows' not in r['Kernel_Name']:
            continue
        name = r['Kernel_Name'].split('(')[0].replace('void dexct::', '')
        last[(name, r['Counter_Name'])] = float(r['Counter_Value'])       # the last dispatch of each kernel wins
names = []
for (n, c) in last:
    if n not in names:
        names.append(n)
for n in names:
    print(n)
    for (m, c), v in last.items():
        if m == n:
            print('   %-44s %.5g' % (c, v))
PY
cat $OUT/g1.txt | grep rows4
