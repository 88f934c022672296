"""configs[2]: 2^32-sample capture, 255-tap windowed sinc, unknown-remote1 -- front-end time, matrix-core against packed-VALU form.

    python tools/config2.py [log2 samples] [valu]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ookiedokie_amd as ok            # noqa: E402
from tests.helpers import golden_path  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 1 << lg
k = np.arange(255) - 127
h = np.sinc(k / 32.0) * np.hamming(255)
h = h / h.sum()
path = os.path.join(tempfile.mkdtemp(), "sinc255.json")
with open(path, "w") as f:
    json.dump({"filter": {"stages": [{"decimation": 1, "taps": list(h)}]}}, f)
flt = ok.Filter.load(path)
dev = ok.Device.load(golden_path("devices", "unknown-remote1"), 3000000)
syn = ok.Synth(dev, n, seed=0xC2, sample_rate=3000000)
cap = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
syn.fill_device(cap.data_ptr())
torch.cuda.synchronize()
out = {}
msgs = {}
for form, kw in (("mfma", {}), ("valu", {"fir_valu": True})):
    if len(sys.argv) > 2 and sys.argv[2] != form:
        continue
    for quiet in (True, False):
        rx = ok.Receiver(flt, dev, max_samples=n, quiet_skip=quiet, **kw)
        rx.process_device(cap.data_ptr(), n)
        t0 = time.perf_counter()
        for _ in range(3):
            rx.process_device(cap.data_ptr(), n)
        dt = (time.perf_counter() - t0) / 3
        st = rx.raw_stats()
        res = rx.result()
        msgs[(form, quiet)] = (list(res.msg_samples), res.payloads.copy())
        out["%s_%s" % (form, "quiet_skip" if quiet else "every_window")] = dict(
            ms_per_step=round(dt * 1e3, 3), front_ms=round(float(st.fir_kernel_ms), 3), gsamples_per_s=round(n / dt / 1e9, 1),
            messages=int(st.num_messages), edges=int(st.num_edges), recomputes=int(st.guard_recomputes), fsm_path=int(st.fsm_path))
        rx.close()
ref = None
for key, v in msgs.items():
    if ref is None:
        ref = v
    assert v[0] == ref[0] and (v[1] == ref[1]).all(), "forms disagree: %s" % (key,)
print(json.dumps(out))
