#!/bin/bash
# per-kernel stats (rocprofv3 --kernel-trace --stats) of a bench command:  tools/kstats.sh <tag> [bench args]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r03}/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-records "$@" > $OUT/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-60s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"].split("(")[0].replace("void ookd::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
