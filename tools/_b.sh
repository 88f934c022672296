cd /root/repo
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["device_ms_per_step"], round(d["device_ms_per_step"]-d["roofline"]["avg_kernel_ms"],4))'
for rep in 1 2 3; do for c in 1 2; do
    echo -n "ctx $c: "; timeout -k 10 120 python bench.py --no-cpu-baseline --steps 100 --contexts $c 2>/dev/null | python -c "$P"
done; done
