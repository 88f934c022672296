#!/bin/bash
# SQ counters of the front-end kernel (run through gpurun):  tools/sq_counters.sh <tag> [bench args...]
# Two PMC passes (8 SQ slots each), kernel trace only beside them.
set -o pipefail
TAG=${1:-sq}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r03}/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/p1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --contexts 1 --no-sub-records "$@" > $OUT/p1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/p2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --contexts 1 --no-sub-records "$@" > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
out = "$OUT"
for p in ("p1", "p2"):
    files = glob.glob(out + "/" + p + "/*/*counter_collection.csv")
    if not files:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "fir1" not in k and "nofir" not in k and "fir2" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    with open(out + "/" + p + "_summary.txt", "w") as w:
        for k in acc:
            nd = len(seen[k])
            w.write("# %s: %d dispatches, per dispatch:\n" % (k, nd))
            for c, v in sorted(acc[k].items()):
                w.write("%s %.0f\n" % (c, v / nd))
    print(open(out + "/" + p + "_summary.txt").read())
PY
