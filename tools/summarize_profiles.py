#!/usr/bin/env python3
"""Turns gpurun_out/prof_<round>/ (tools/collect_profiles.sh) into the tracked
files under profiles/:

    python tools/summarize_profiles.py r01

  <round>_kernel_stats.csv       rocprofv3 --kernel-trace --stats summary of the default bench command
  <round>_kernel_stats_one_context.csv   the same with --contexts 1 (every kernel undisturbed)
  traffic.json                   HBM bytes per launch of the front-end kernel
                                 (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes)
  <round>_bench_line*.json       the bench lines of the same build
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "fir1_mfma_kernel"
SAMPLES_PER_LAUNCH = 1 << 29          # the front end of a 2^32-sample capture goes out as 8 grid launches
ALGO_BYTES = 4.125 * SAMPLES_PER_LAUNCH


def counter_avg(dirname, counter):
    vals = []
    # gpurun_out/ accumulates the files of earlier collections: only the newest one counts
    files = sorted(glob.glob(os.path.join(dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) > (1 << 22):
                    vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
    dst = os.path.join(ROOT, "profiles")
    stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(dst, rnd + "_kernel_stats.csv"))
    fetch, nf = counter_avg(os.path.join(src, "fetch"), "FETCH_SIZE")
    write, nw = counter_avg(os.path.join(src, "write"), "WRITE_SIZE")
    if fetch is not None and write is not None:
        hbm = (2.0 * fetch + write) * 1024.0
        with open(os.path.join(dst, "traffic.json"), "w") as f:
            json.dump({
                "kernel": "ookd::" + KERNEL + "<4>",
                "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python "
                           "bench.py --steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --no-sub-records (two separate passes, "
                           "tools/collect_profiles.sh)",
                "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write, "launches": [nf, nw],
                "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane streaming reads "
                              "-> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
                "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": ALGO_BYTES,
                "samples_per_launch": SAMPLES_PER_LAUNCH,
            }, f, indent=1)
        print("traffic: %.4f GB per launch (algorithmic %.4f GB)" % (hbm / 1e9, ALGO_BYTES / 1e9))
    stats1 = sorted(glob.glob(os.path.join(src, "stats1", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if stats1:
        shutil.copy(stats1[-1], os.path.join(dst, rnd + "_kernel_stats_one_context.csv"))
    # SQ counters of the front-end kernel with every window loud (tools/sq_counters.sh worst --no-quiet-skip) and of configs[2]
    for tag, out in (("worst", rnd + "_fir1_sq_counters.txt"), ("config2", rnd + "_fir1_sq_counters_255taps.txt")):
        parts = []
        for pp in ("p1", "p2"):
            fsum = os.path.join(ROOT, "gpurun_out", rnd, tag, pp + "_summary.txt")
            if os.path.exists(fsum):
                parts.append(open(fsum).read())
        if parts:
            with open(os.path.join(dst, out), "w") as f:
                f.write("# rocprofv3 --pmc <8 SQ counters> --kernel-trace, two passes (tools/sq_counters.sh / sq_counters_cmd.sh); per dispatch of the\n"
                        "# front-end kernel (2^29 samples = 524 288 wave tiles of 1024 outputs); WAVE / WAIT / ACTIVE in quad-cycles\n")
                f.write("".join(parts))
    for name in ("bench_line", "bench_line_dec4", "bench_line_batch", "bench_line_1GiB"):
        p = os.path.join(src, name + ".json")
        if os.path.exists(p):
            with open(p) as f:
                lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
            if lines:
                with open(os.path.join(dst, "%s_%s.json" % (rnd, name)), "w") as f:
                    f.write(lines[-1] + "\n")
                j = json.loads(lines[-1])
                print(name, j["value"], j["roofline"]["frac"], j["roofline"]["avg_kernel_ms"])


if __name__ == "__main__":
    main()
