#!/bin/bash
# A/B of front-end settings through bench.py:  tools/front_ab.sh <filter> "ENV=.." "ENV=.." ...   (on the GPU box)
mkdir -p gpurun_out/r03
FILT=$1; shift
for cfg in "$@"; do
  env OOKD_DEVELOPER=1 $cfg timeout -k 10 300 python bench.py --filter $FILT --steps 24 --warmup 4 --no-cpu-baseline > gpurun_out/r03/dbg.json 2> gpurun_out/r03/dbg.err || { tail -5 gpurun_out/r03/dbg.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03/dbg.json"))
w=d.get("worst_case") or {}
print("$FILT $cfg", "3ctx ms", d["ms_per_step"], "front", round(d["roofline"]["avg_kernel_ms"]*d["roofline"]["launches_per_step"],3), "| 1ctx", d["single_context"]["ms_per_step"], d["single_context"]["kernel_ms"], "| worst", w.get("ms_per_step"), w.get("kernel_ms"), flush=True)
PY
done
