"""Which way the scan found the leaves' entry states (stats.scan_entry_form: 1 walk from synchronising spans,
2 composed block tables), on the bench capture and on a few hostile ones; both must decode the same.

    python tools/sync_check.py [log2 samples]
"""
import os
import sys

os.environ["OOKD_DEVELOPER"] = "1"
os.environ["OOKD_SYNC_MIN_EDGES"] = "0"      # the walk whatever the edge count (default: from 20 000 edges on)

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ookiedokie_amd as ok            # noqa: E402
from tests.helpers import golden_path  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << lg
flt = ok.Filter.load(golden_path("filters", "fs32_fs4"))
dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), 3000000)
d = torch.empty(2 * n, dtype=torch.int16, device="cuda")
for name, kw, cap in (("bench", {}, 0), ("no_glitches", dict(glitch_every=0), 0), ("glitch_every_5", dict(glitch_every=5), 0),
                      ("short_gaps", dict(gap_us=(4000, 6000)), 0), ("long_gaps", dict(gap_us=(17000, 30000)), 0),
                      ("noise_900", dict(noise=900), n // 4)):
    syn = ok.Synth(dev, n, seed=5, sample_rate=3000000, **kw)
    syn.fill_device(d.data_ptr())
    out = {}
    for tables in (False, True):
        rx = ok.Receiver(flt, dev, max_samples=n, scan_tables=tables, edge_capacity=cap)
        for _ in range(3):
            rx.process_device(d.data_ptr(), n)
        res = rx.result()
        st = rx.raw_stats()
        errs, nerr = rx.errors()
        out[tables] = (list(res.msg_samples), res.payloads.tobytes(), int(nerr), list(errs[:64]))
        print(name, "tables" if tables else "default", "form", st.scan_entry_form, "path", st.fsm_path, "msgs", st.num_messages,
              "errors", st.num_errors, "edges", st.num_edges, "device ms %.3f" % st.total_device_ms,
              "front %.3f" % st.fir_kernel_ms, "chain %.3f" % (st.total_device_ms - st.fir_kernel_ms), flush=True)
        rx.close()
    assert out[False] == out[True], name
# an edge list that overflows by far: an error, not a fault
m = 1 << 24
syn = ok.Synth(dev, m, seed=5, sample_rate=3000000, noise=900)
syn.fill_device(d.data_ptr())
for rounds in (False, True):
    rx = ok.Receiver(flt, dev, max_samples=m, fsm_rounds=rounds)
    try:
        rx.process_device(d.data_ptr(), m)
        print("overflow: no error?!")
    except ok.OokdError as e:
        print("overflow (rounds=%s): %s" % (rounds, e))
    rx.close()
print("same results")
