#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/st
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --contexts 1 --no-cpu-baseline --no-sub-records > $OUT/log 2>&1
python - <<'PY'
import csv,glob,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/st'
f=glob.glob(out+'/t/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void ','').replace('ookd::','')[:40]
    print("%-42s calls %5s avg %10.1f us" % (n, r['Calls'], float(r['AverageNs'])/1e3))
PY
