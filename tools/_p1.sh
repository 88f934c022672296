set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_b1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "default rc=$?"
timeout -k 10 300 python bench.py --workload batch --no-cpu-baseline > $OUT/bench_batch.json 2> $OUT/batch.err; echo "batch rc=$?"
timeout -k 10 100 python bench.py --gpus 2 --steps 3 > $OUT/gpus2.out 2> $OUT/gpus2.err; echo "gpus2 on 1-GPU box rc=$? (want non-zero)"; tail -1 $OUT/gpus2.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload sharded --samples 268435456 --steps 5 --warmup 2 > $OUT/sharded2.json 2> $OUT/sharded2.err; echo "sharded 2 ranks (gloo, one GPU) rc=$?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --workload batch --steps 5 --warmup 2 --contexts 1 > $OUT/batch2.json 2> $OUT/batch2.err; echo "batch 2 ranks rc=$?"
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02_b1/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-20s %9.1f Ms/s %.4f ms/step n_gpus %d roof %.3f single %s worst %s cpu %s" % (os.path.basename(f), d['value'], d['ms_per_step'], d['n_gpus'], d['roofline']['frac'], (d.get('single_context') or {}).get('ms_per_step'), (d.get('worst_case') or {}).get('kernel_ms'), (d.get('cpu_baseline') or {}).get('value')))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
tail -3 $OUT/bench.err $OUT/sharded2.err | cut -c1-300
