#!/bin/bash
mkdir -p gpurun_out/r03
for c in "$@"; do
  timeout -k 10 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-sub-records --contexts $c > gpurun_out/r03/dbg.json 2> gpurun_out/r03/dbg.err || { tail -5 gpurun_out/r03/dbg.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03/dbg.json"))
print("contexts $c", "ms", d["ms_per_step"], "front", round(d["roofline"]["avg_kernel_ms"]*8,3), "dev_ms", d["device_ms_per_step"], flush=True)
PY
done
