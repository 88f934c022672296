#!/usr/bin/env python3
"""Front-end kernel time on three captures (all quiet, all loud, the bench capture), grid and streaming form.
    python tools/front_diag.py [log2 samples]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ookiedokie_amd as ok

def golden(kind, name):
    return os.path.join(ROOT, "tests", "golden", kind, name + ".json")

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
flt = ok.Filter.load(golden("filters", "fs32_fs4"))
dev = ok.Device.load(golden("devices", "p3l-nexa2012"), 3000000)
caps = {}
caps["quiet"] = torch.zeros(2 * n + 64, dtype=torch.int16, device="cuda")
caps["loud"] = torch.full((2 * n + 64,), 1500, dtype=torch.int16, device="cuda")
b = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
ok.Synth(dev, n, seed=0x00C0FFEE + 2, sample_rate=3000000).fill_device(b.data_ptr())
caps["bench"] = b
torch.cuda.synchronize()
for form in ("grid", "stream"):
    rx = ok.Receiver(flt, None, max_samples=n, front_grid=(form == "grid"))
    for name, c in caps.items():
        ts = []
        for _ in range(6):
            rx.process_device(c.data_ptr(), n)
            ts.append(rx.stats()["fir_kernel_ms"])
        t = min(ts[1:])
        print("%-6s %-5s %.4f ms  %.2f TB/s (4.125 B/sample)" % (form, name, t, 4.125 * n / t / 1e9), flush=True)
    rx.close()
