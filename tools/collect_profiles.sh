#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box (run through gpurun):
#   tools/collect_profiles.sh r02
# Outputs under gpurun_out/prof_<round>/; tools/summarize_profiles.py turns them into profiles/.
# (PMC counters in their own passes, never beside a trace domain other than the kernel trace.)
set -o pipefail
ROUND=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sub-records > $OUT/stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --contexts 1 --no-cpu-baseline --no-sub-records > $OUT/stats1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --no-sub-records > $OUT/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --no-sub-records > $OUT/write.log 2>&1 &&
cd $GRAFT_REPO_ROOT && timeout -k 10 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -1 $OUT/bench_line.json | cut -c1-200
