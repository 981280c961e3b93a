#!/bin/bash
# The cone beam with 12 table rows (4 material-group passes of cone_cols_kernel + one detection pass, round 6) under rocprofv3:
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes.  usage: tools/profile_cone_groups.sh <tag> [n_mat]
set -o pipefail
TAG=${1:-r06_cone_groups}; NMAT=${2:-12}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
export CONE_GROUPS_ONLY=groups
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $REPO/tools/probes/cone_groups.py $NMAT > $OUT/run_$name.log 2> $OUT/run_$name.err
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $OUT/run_$name.err; exit $rc; fi
}
run stats --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
cd $REPO && python3 - <<PY
import csv, glob, os
src = 'gpurun_out/prof_$TAG'
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(newest(os.path.join(src, 'stats', '*', '*_kernel_stats.csv')))))
def pmc(which):
    out = {}
    for r in csv.DictReader(open(newest(os.path.join(src, f'pmc_{which}', '*', '*_counter_collection.csv')))):
        out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']) * 1024.0)
    return {k: sum(v) / len(v) for k, v in out.items()}
f, w = pmc('fetch'), pmc('write')
print(open(os.path.join(src, 'run_stats.log')).read().strip())
print('| kernel | calls | avg ms | FETCH_SIZE GB (raw) | WRITE_SIZE GB | per dispatch |')
print('|---|---|---|---|---|---|')
for r in rows[:8]:
    k = r['Name']
    print(f'| \`{k[:72]}\` | {r["Calls"]} | {float(r["AverageNs"]) / 1e6:.3f} | {f.get(k, 0) / 1e9:.3f} | {w.get(k, 0) / 1e9:.3f} | |')
PY
