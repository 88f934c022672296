import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ookiedokie_amd as ok
from tests.helpers import golden_path
n = 1 << 28
k = np.arange(255) - 127
h = np.sinc(k / 32.0) * np.hamming(255); h = h / h.sum()
print("sum|h|", np.abs(h.astype(np.float32)).sum())
with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
    json.dump({"filter": {"stages": [{"decimation": 1, "taps": list(h)}]}}, f); fpath = f.name
flt = ok.Filter.load(fpath)
for devname in ("unknown-remote1", "p3l-nexa2012"):
    dev = ok.Device.load(golden_path("devices", devname), 3000000)
    syn = ok.Synth(dev, n, seed=0xC2, sample_rate=3000000)
    cap = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
    syn.fill_device(cap.data_ptr()); torch.cuda.synchronize()
    for qs in (True, False):
        rx = ok.Receiver(flt, dev, max_samples=n, count_quiet=qs, quiet_skip=qs)
        for _ in range(3):
            st = rx.rx_device(cap.data_ptr(), n).stats
        qf = st["quiet_waves"] / max(1, st["total_waves"])
        loud = (1 - qf) * n
        print(devname, "quiet_skip", qs, "fir_ms %.3f" % st["fir_kernel_ms"], "quiet frac %.3f" % qf,
              "TFLOP/s on loud windows %.1f" % (loud * 1020 / (st["fir_kernel_ms"] * 1e-3) / 1e12))
        rx.close()
    del cap
