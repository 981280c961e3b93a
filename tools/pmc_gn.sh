#!/bin/bash
# Counter passes for the Newton kernel (each its own rocprofv3 run, counters only + kernel trace):
# effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), issue/wait split, instruction mix.
# usage: tools/pmc_gn.sh <tag>         (DEXCT_GN_FULL_LOOP=1 in the environment: every iteration executed)
TAG=${1:-gn}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --skip-single-row --skip-gn-full-loop --skip-quadrature"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk -- python3 $REPO/bench.py $ARGS "$@" > $OUT/clk.json 2> $OUT/clk.err
rc=$?; echo "clk rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/bench.py $ARGS "$@" > $OUT/a.json 2> $OUT/a.err
rc=$?; echo "a rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/bench.py $ARGS "$@" > $OUT/b.json 2> $OUT/b.err
rc=$?; echo "b rc=$rc"; if [ $rc -ne 0 ]; then echo "rocprofv3 pass failed: stopping (no summary from partial CSVs)"; exit $rc; fi
python3 - <<PY
import csv, glob, collections
dur = {}
for f in glob.glob('$OUT/clk/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if ('gn_' in r['Kernel_Name'] and 'tables' not in r['Kernel_Name']) or 'rows4' in r['Kernel_Name']:
            dur.setdefault(r['Kernel_Name'][:40], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in dur.items(): print('duration_ns', k, v)
for sub in ('clk', 'a', 'b'):
    for f in glob.glob('$OUT/%s/*/*_counter_collection.csv' % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if ('gn_' in r['Kernel_Name'] and 'tables' not in r['Kernel_Name']) or 'rows4' in r['Kernel_Name']:
                agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k in sorted(agg): print(k, ['%.5g' % x for x in agg[k]])
PY
