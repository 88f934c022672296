#!/usr/bin/env python3
"""Cost of the state machine's fallback on the bench capture (GPU box):
scan form, round form, and scan form refusing the capture because of one
event it cannot represent (a bit gap outside both windows: no trigger fires
on that edge, the counter runs on).  Prints one JSON line.

    python tools/fallback_cost.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ookiedokie_amd as ok                         # noqa: E402
from tests.helpers import golden_path              # noqa: E402

N = 1 << 28
RATE = 3000000


def timed(rx, ptr, n, reps=10):
    for _ in range(3):
        res = rx.rx_device(ptr, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = rx.rx_device(ptr, n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, res


def main():
    flt = ok.Filter.load(golden_path("filters", "fs32_fs4"))
    dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), RATE)
    syn = ok.Synth(dev, N, seed=0x00C0FFEE + 2, sample_rate=RATE)
    cap = torch.empty(2 * N + 64, dtype=torch.int16, device="cuda")
    syn.fill_device(cap.data_ptr(), hip_device=0)
    torch.cuda.synchronize()
    out = {}
    rx = ok.Receiver(flt, dev, max_samples=N)
    ms, res = timed(rx, cap.data_ptr(), N)
    out["scan_ms"] = round(ms, 3)
    out["scan_path"] = res.stats["fsm_path"]
    out["messages"] = len(res.msg_samples)
    rx.close()
    rx = ok.Receiver(flt, dev, max_samples=N, fsm_rounds=True)
    ms, res2 = timed(rx, cap.data_ptr(), N)
    out["rounds_ms"] = round(ms, 3)
    out["rounds_iterations"] = res2.stats["fsm_iterations"]
    assert list(res2.msg_samples) == list(res.msg_samples)
    rx.close()
    # one unrepresentable event in the first 100 000 samples
    head = np.zeros(2 * 100000, dtype=np.int16)
    pos = 1000
    for i, run in enumerate([1500, 26100, 1500, 8500, 1500, 6000, 1500]):
        if i % 2 == 0:
            head[2 * pos:2 * (pos + run):2] = 1945
        pos += run
    cap[:head.size] = torch.from_numpy(head).cuda()
    torch.cuda.synchronize()
    rx = ok.Receiver(flt, dev, max_samples=N)
    ms, res3 = timed(rx, cap.data_ptr(), N)
    out["scan_refusing_ms"] = round(ms, 3)
    out["scan_refusing_path"] = res3.stats["fsm_path"]
    out["scan_refusing_reason"] = res3.stats["fsm_fallback_reason"]
    rx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
