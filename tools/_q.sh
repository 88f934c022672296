cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() { name=$1; ctx=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --contexts $ctx --steps 12 --warmup 3 --no-cpu-baseline --no-sub-records | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['roofline']['front_end_ms_per_step'], d['device_ms_per_step'], d['config']['fsm_path'])"; }
run tables_c1 1 OOKD_EMIT_TABLES=1
run entry_c1 1 X=1
run tables_c3 3 OOKD_EMIT_TABLES=1
run entry_c3 3 X=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_dw
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --contexts 1 --no-cpu-baseline --no-sub-records > $OUT/stats3.log 2>&1
cat $OUT/stats3/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-100 | head -12
