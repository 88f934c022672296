#!/bin/bash
mkdir -p gpurun_out/r03
for cfg in "$@"; do
  env OOKD_DEVELOPER=1 $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --contexts 1 --no-sub-records > gpurun_out/r03/dbg.json 2> gpurun_out/r03/dbg.err || { tail -5 gpurun_out/r03/dbg.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03/dbg.json"))
print("$cfg", "ms", d["ms_per_step"], "kernel", round(d["roofline"]["avg_kernel_ms"]*8,3), flush=True)
PY
done
