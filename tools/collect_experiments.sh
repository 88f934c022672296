#!/bin/bash
# The round-2 design experiments, as files (run through gpurun; copy what matters into profiles/):
#   tools/collect_experiments.sh r02
set -o pipefail
ROUND=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/exp_$ROUND
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -w tools/stream_bw.hip -o tools/stream_bw &&
hipcc --offload-arch=gfx950 -O3 -w -mllvm -amdgpu-atomic-optimizer-strategy=None tools/stream_bw2.hip -o tools/stream_bw2 &&
timeout -k 5 200 ./tools/stream_bw 8 > $OUT/stream_bw.txt 2>&1 &&
timeout -k 5 200 ./tools/stream_bw2 4 > $OUT/stream_bw2.txt 2>&1 &&
timeout -k 5 300 python tools/cu_mask_probe.py 28 > $OUT/cu_mask_probe.txt 2>&1 &&
( echo "# hardware-dispatched grid form"; timeout -k 5 200 python tools/front_diag.py 28 | grep grid;
  for W in 12 16; do echo "# streaming (persistent) form, $W waves per CU"; OOKD_DEVELOPER=1 OOKD_FRONT_STREAM=1 OOKD_STREAM_WAVES=$W timeout -k 5 200 python tools/front_diag.py 28 | grep stream; done ) > $OUT/front_forms.txt 2>&1 &&
for V in "whole OOKD_NO_PIPELINE=1" "pipelined_1GiB_chunks OOKD_PIPELINE=1"; do set -- $V; env $2 timeout -k 10 300 python bench.py --contexts 1 --steps 8 --warmup 2 --no-cpu-baseline --no-sub-records > $OUT/bench_$1.json 2> $OUT/bench_$1.err; done
cd /tmp && export TMPDIR=/tmp
OOKD_PIPELINE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/pipe_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-sub-records > $OUT/pipe_trace.log 2>&1
python - <<'PY'
import csv,glob,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/exp_'+(os.environ.get('ROUND') or 'r02')
f=glob.glob(out+'/pipe_trace/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'clear_tiles' in r['Kernel_Name']]
rows=rows[idx[-1]:]
t0=int(rows[0]['Start_Timestamp'])
with open(out+'/pipeline_timeline.txt','w') as w:
    w.write("# one pipelined step (16 GiB capture, 1 GiB chunks): kernel start / end in us since the step began\n")
    for r in rows[:90]:
        n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('ookd::','')
        w.write("%9.1f %9.1f  %s\n" % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3,n[:50]))
PY
ls $OUT
