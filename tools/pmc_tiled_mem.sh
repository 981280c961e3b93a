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
  # made rocprofv3 abort in round 2: gpurun_out/pmc_m1/g3.err reads "rocprofiler_create_counter_config ... error code 38:
  # Request exceeds the capabilities of the hardware to collect" - FOUR counters of ONE block (TA) in one pass exceed that
  # block's counter slots; it is not a fault of the TA block or of the kernel.  If TA stalls are wanted, request them one
  # or two per pass, like the TCP groups above stay within their block's slots.)
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/tools/bench_tiled.py --reps 1 "$@" > $OUT/g$i.txt 2> $OUT/g$i.err
  rc=$?
  echo "group $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $OUT/g$i.err; exit $rc; fi      # never parse the partial CSVs of a failed pass
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
