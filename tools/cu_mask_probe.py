#!/usr/bin/env python3
"""Front-end kernel time when its stream may only use part of the chip (hipExtStreamCreateWithCUMask).
    python tools/cu_mask_probe.py [log2 samples]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ookiedokie_amd as ok

def golden(kind, name):
    return os.path.join(ROOT, "tests", "golden", kind, name + ".json")

hip = C.CDLL("libamdhip64.so")
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
flt = ok.Filter.load(golden("filters", "fs32_fs4"))
dev = ok.Device.load(golden("devices", "p3l-nexa2012"), 3000000)
cap = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
ok.Synth(dev, n, seed=0x00C0FFEE + 2, sample_rate=3000000).fill_device(cap.data_ptr())
torch.cuda.synchronize()
for pattern, name in ((0xFFFFFFFF, "256 CUs (all)"), (0xFFFFFFFE, "248 (31 of 32)"), (0xFFFFFFFC, "240"), (0xFFFFFFF0, "224"),
                      (0xFFFFFF00, "192"), (0xFFFF0000, "128"), (0x77777777, "192 (3 of every 4)"), (0x55555555, "128 (every other)")):
    mask = (C.c_uint32 * 8)(*([pattern] * 8))
    stream = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, mask)
    if rc != 0:
        print("hipExtStreamCreateWithCUMask failed", rc)
        break
    rx = ok.Receiver(flt, dev, max_samples=n, stream=stream.value)
    ts, tt = [], []
    for _ in range(6):
        rx.process_device(cap.data_ptr(), n)
        st = rx.stats()
        ts.append(st["fir_kernel_ms"])
        tt.append(st["total_device_ms"])
    print("%-22s front end %.4f ms, whole step on device %.4f ms" % (name, min(ts[1:]), min(tt[1:])), flush=True)
    rx.close()
    hip.hipStreamDestroy(stream)
