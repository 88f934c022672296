#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box (run through gpurun):
#   tools/collect_profiles.sh r01
# Outputs under gpurun_out/prof_<round>/; tools/summarize_profiles.py turns them into profiles/.
set -o pipefail
ROUND=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --contexts 1 --no-cpu-baseline > $OUT/stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1 &&
cd $GRAFT_REPO_ROOT && timeout -k 10 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err &&
timeout -k 10 300 python bench.py --no-quiet-skip --no-cpu-baseline > $OUT/bench_line_no_quiet_skip.json 2>> $OUT/bench.err &&
timeout -k 10 300 python bench.py --filter fs128_fs16_dec4 --no-cpu-baseline > $OUT/bench_line_dec4.json 2>> $OUT/bench.err &&
timeout -k 10 300 python bench.py --contexts 1 --no-cpu-baseline > $OUT/bench_line_one_context.json 2>> $OUT/bench.err &&
timeout -k 10 300 python bench.py --contexts 2 --no-cpu-baseline > $OUT/bench_line_two_contexts.json 2>> $OUT/bench.err
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench_line.json | cut -c1-300
