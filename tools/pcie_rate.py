#!/usr/bin/env python3
"""PCIe-inclusive rates of the rx path (NOT the bench metric: bench.py times
the path with the capture already resident in HBM).  Prints one JSON line:

  host_buffer : ookd_rx_process_host over a pageable numpy capture
                (pinned double-buffered H2D + the whole hot path)
  file        : sdr_hip_file_capture of a .sc16q11 file in the page cache
                (fread -> pinned -> HBM) + ookd_rx_process_device

    python tools/pcie_rate.py [--samples N]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1 << 27)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import ookiedokie_amd as ok

    n = args.samples
    g = lambda kind, name: os.path.join(ROOT, "tests", "golden", kind, name + ".json")
    flt = ok.Filter.load(g("filters", "fs32_fs4"))
    dev = ok.Device.load(g("devices", "p3l-nexa2012"), 3_000_000)
    iq = ok.Synth(dev, n, seed=7, sample_rate=3_000_000).fill_host()
    rx = ok.Receiver(flt, dev, max_samples=n, threshold=0.1, samples_per_buffer=8192)
    rx.rx(iq)
    t = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        res = rx.rx(iq)
        t.append(time.perf_counter() - t0)
    host_s = min(t)

    with tempfile.NamedTemporaryFile(suffix=".sc16q11", dir="/dev/shm", delete=False) as f:
        path = f.name
    try:
        iq.tofile(path)
        t = []
        for _ in range(args.reps):
            be = ok.HipFileBackend(path, rx=True, samples_per_buffer=8192)
            t0 = time.perf_counter()
            ptr, cnt = be.capture()
            t1 = time.perf_counter()
            res2 = rx.rx_device(ptr, cnt)
            t2 = time.perf_counter()
            t.append((t2 - t0, t1 - t0))
            be.close()
        file_s, ingest_s = min(t)
    finally:
        os.unlink(path)
    assert list(res2.msg_samples) == list(res.msg_samples)
    print(json.dumps({
        "samples": n, "messages": int(len(res.msg_samples)),
        "host_buffer": {"seconds": round(host_s, 4), "Msamples_per_s": round(n / host_s / 1e6, 1),
                        "GB_per_s": round(4 * n / host_s / 1e9, 2)},
        "file": {"seconds": round(file_s, 4), "ingest_seconds": round(ingest_s, 4),
                 "Msamples_per_s": round(n / file_s / 1e6, 1), "GB_per_s": round(4 * n / file_s / 1e9, 2)},
    }))


if __name__ == "__main__":
    main()
