"""Matrix-core FIR form (csrc/fir_mfma.hip) against the exact packed-VALU kernel on loud random captures.

    python tools/mfma_check.py [log2 samples]

For each of fs32_fs4 / a 64-tap / a 255-tap windowed sinc and each amplitude class (nominal: |x| <= 2047,
wide: any int16), on a capture where EVERY window is loud:
  * bits of the default (matrix-core) front end == bits of OOKD_RX_EXACT_FIR (reference order) -- must hold;
  * float output against a float64 evaluation of the real sum on a slice: max |err| / (sum|h| max|x|);
  * guard-band recomputes;
  * kernel time of both forms and of the fused packed-VALU form (hipEvents of the library's own front-end stamps).
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ookiedokie_amd as ok            # noqa: E402
from tests.helpers import golden_path  # noqa: E402


def sinc_filter(ntaps, tmp):
    n = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(n / 32.0) * np.hamming(ntaps)
    h = h / h.sum()
    p = os.path.join(tmp, "sinc%d.json" % ntaps)
    with open(p, "w") as f:
        json.dump({"filter": {"stages": [{"decimation": 1, "taps": list(h)}]}}, f)
    return p


def capture(n, wide, seed):
    rng = np.random.default_rng(seed)
    amp = 30000 if wide else 2047
    # slow envelope crossing the threshold region often + noise: plenty of samples near |y| = 0.1
    t = np.arange(n)
    env = 0.1 * 2048 * (1.0 + 0.5 * np.sin(2 * np.pi * t / 5000.0)) * (16 if wide else 1)
    ph = rng.uniform(0, 2 * np.pi)
    i = env * np.cos(ph) + rng.integers(-amp // 8, amp // 8 + 1, size=n)
    q = env * np.sin(ph) + rng.integers(-amp // 8, amp // 8 + 1, size=n)
    if wide:
        # sprinkle extreme values
        idx = rng.integers(0, n, size=n // 50)
        i[idx] = rng.choice([-32768, 32767, -2049, 2048], size=idx.size)
    iq = np.empty(2 * n, dtype=np.int16)
    iq[0::2] = np.clip(i, -amp if not wide else -32768, amp if not wide else 32767).astype(np.int16)
    iq[1::2] = np.clip(q, -amp if not wide else -32768, amp if not wide else 32767).astype(np.int16)
    return iq


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n = 1 << lg
    tmp = tempfile.mkdtemp()
    filters = [("fs32_fs4", golden_path("filters", "fs32_fs4")), ("sinc64", sinc_filter(64, tmp)),
               ("sinc255", sinc_filter(255, tmp))]
    out = []
    for name, path in filters:
        flt = ok.Filter.load(path)
        with open(path) as f:
            taps = np.array(json.load(f)["filter"]["stages"][0]["taps"], dtype=np.float64).astype(np.float32)
        for wide in (False, True):
            iq = capture(n, wide, 7 + int(wide))
            d = torch.from_numpy(iq).cuda()
            res = {}
            for form, kw in (("mfma", {}), ("valu_fma", {"fir_valu": True}), ("exact", {"exact_fir": True})):
                rx = ok.Receiver(flt, None, max_samples=n, quiet_skip=False, keep_fir=True, edge_capacity=n, **kw)
                for _ in range(3):
                    rx.process_device(d.data_ptr(), n)
                st = rx.raw_stats()
                res[form] = dict(bits=rx.bits().copy(), y=rx.fir_output().copy(),
                                 ms=float(st.fir_kernel_ms), redo=int(st.guard_recomputes))
                rx.close()
                # timing without the float output
                rx = ok.Receiver(flt, None, max_samples=n, quiet_skip=False, edge_capacity=n, **kw)
                for _ in range(3):
                    rx.process_device(d.data_ptr(), n)
                res[form]["ms_bits_only"] = float(rx.raw_stats().fir_kernel_ms)
                rx.close()
            same = bool((res["mfma"]["bits"] == res["exact"]["bits"]).all())
            same_v = bool((res["valu_fma"]["bits"] == res["exact"]["bits"]).all())
            # float64 reference on a slice
            m = min(n, 1 << 16)
            x = iq[: 2 * m].astype(np.float64).reshape(-1, 2) / 2048.0
            yr = np.convolve(x[:, 0], taps.astype(np.float64))[:m]
            yi = np.convolve(x[:, 1], taps.astype(np.float64))[:m]
            scale = float(np.abs(taps).sum()) * float(np.abs(iq).max()) / 2048.0
            errs = {}
            for form in res:
                y = res[form]["y"].reshape(-1, 2)[:m].astype(np.float64)
                errs[form] = float(max(np.abs(y[:, 0] - yr).max(), np.abs(y[:, 1] - yi).max()) / scale)
            rec = dict(filter=name, wide=wide, samples=n, bits_equal_exact=same, valu_bits_equal_exact=same_v,
                       ones=int(res["exact"]["bits"].sum()),
                       max_err_over_scale=errs,
                       recomputes={k: v["redo"] for k, v in res.items()},
                       front_ms={k: round(v["ms_bits_only"], 4) for k, v in res.items()},
                       gsamples_per_s={k: round(n / v["ms_bits_only"] / 1e6, 1) for k, v in res.items()})
            print(json.dumps(rec), flush=True)
            out.append(rec)
            assert same, "matrix-core bits differ from the exact kernel"
    print("all equal")


if __name__ == "__main__":
    main()
