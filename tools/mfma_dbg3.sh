#!/bin/bash
# like mfma_dbg.sh but with three contexts in flight (the default bench shape)
mkdir -p gpurun_out/r03
for cfg in "$@"; do
  env OOKD_DEVELOPER=1 $cfg timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-sub-records > gpurun_out/r03/dbg.json 2> gpurun_out/r03/dbg.err || { tail -5 gpurun_out/r03/dbg.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03/dbg.json"))
print("$cfg", "3ctx ms", d["ms_per_step"], "front", round(d["roofline"]["avg_kernel_ms"]*8,3), "dev_ms", d["device_ms_per_step"], flush=True)
PY
done
