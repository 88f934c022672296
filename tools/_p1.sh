set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_p8
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { name=$1; ctx=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --samples 4294967296 --contexts $ctx --steps 12 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; }
run c1_auto 1 X=1
run c1_g512 1 OOKD_SCAN_GRID=512
run c1_g768 1 OOKD_SCAN_GRID=768
run c1_g1024 1 OOKD_SCAN_GRID=1024
run c1_g1536 1 OOKD_SCAN_GRID=1536
run c3_auto 3 X=1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02_p8/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-16s %9.1f Ms/s  %.4f ms/step  fir %.4f ms  dev %.4f" % (os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['device_ms_per_step']))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
