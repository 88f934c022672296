#!/bin/bash
# sweep of FrontParams::mfma_g (wave tiles per workgroup of the matrix-core front end)
mkdir -p gpurun_out/r03
for g in ${@:-2 4 8 16}; do
  OOKD_DEVELOPER=1 OOKD_MFMA_G=$g timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03/sweep_g$g.json 2> gpurun_out/r03/sweep_g$g.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r03/sweep_g$g.json"))
print("G=$g", "3ctx ms", d["ms_per_step"], "kernel", d["roofline"]["avg_kernel_ms"]*8, "| 1ctx", d["single_context"]["ms_per_step"], "kernel", d["single_context"]["kernel_ms"], "| worst", d["worst_case"]["ms_per_step"], "kernel", d["worst_case"]["kernel_ms"], flush=True)
PY
done
