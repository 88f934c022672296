#!/bin/bash
# how many waves per CU does the streaming front end need?  (packed-VALU kernel, LDS padding caps residency)
mkdir -p gpurun_out/r03
for pad in ${@:-0 2560 5120 7680 10240}; do
  OOKD_DEVELOPER=1 OOKD_FIR_VALU=1 OOKD_FIR1_LDS_PAD=$pad timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --contexts 1 --no-sub-records > gpurun_out/r03/occ_$pad.json 2> gpurun_out/r03/occ_$pad.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r03/occ_$pad.json"))
print("pad=$pad", "ms", d["ms_per_step"], "kernel", d["roofline"]["avg_kernel_ms"]*8, flush=True)
PY
done
