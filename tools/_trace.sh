#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/ov_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 3 --contexts 3 --no-cpu-baseline --no-sub-records > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/overlap_cost.py $(ls $OUT/t/*/*kernel_trace.csv | head -1) > $OUT/overlap_cost.txt 2>&1
tail -3 $OUT/trace.log | cut -c1-200
