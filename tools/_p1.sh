set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_p6
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
run() { name=$1; ctx=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --samples 4294967296 --contexts $ctx --steps 12 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; }
run c1_l29 1 OOKD_FRONT_LAUNCH_LOG2=29
run c3_l29 3 OOKD_FRONT_LAUNCH_LOG2=29
run c3_l30 3 OOKD_FRONT_LAUNCH_LOG2=30
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02_p6/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-16s %9.1f Ms/s  %.4f ms/step  fir %.4f ms  dev %.4f" % (os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['device_ms_per_step']))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
