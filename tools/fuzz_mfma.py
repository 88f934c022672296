#!/usr/bin/env python3
"""Randomised differential run of the matrix-core FIR form (csrc/fir_mfma.hip) on the GPU box: random filters
(1..256 taps: windowed sincs, random, sparse, wide dynamic range), random captures (nominal / wide sample range,
quiet and loud stretches, amplitudes hovering around the threshold), random thresholds and lengths; the bits of the
matrix-core form and of the packed-VALU form must be the CPU oracle's, the floats within 1e-5 of the scale.
Not part of the test suite (minutes); prints a JSON summary.

    python tools/fuzz_mfma.py [--seconds 300] [--seed 1]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ookiedokie_amd as ok          # noqa: E402
import oracle as O                   # noqa: E402  (checker)


def random_taps(rng, nmax=256):
    n = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256,
                        int(rng.integers(1, 257))]))
    if nmax < 256:
        n = int(rng.choice([1, 2, nmax - 1, nmax, int(rng.integers(1, nmax + 1))]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        k = np.arange(n) - (n - 1) / 2.0
        h = np.sinc(k / rng.uniform(2.0, 40.0)) * np.hamming(n + 2)[1:-1]
        h = h / (np.abs(h.sum()) + 1e-30)
    elif kind == 1:
        h = rng.normal(0, 1, n)
    elif kind == 2:
        h = np.zeros(n)
        idx = rng.integers(0, n, size=max(1, n // 8))
        h[idx] = rng.normal(0, 1, idx.size)
    elif kind == 3:
        h = rng.normal(0, 1, n) * 2.0 ** rng.integers(-28, 1, size=n)      # wide dynamic range
    else:
        h = np.ones(n) / n
    if not np.any(h):
        h[0] = 1.0
    scale = float(rng.choice([1.0, 1.0, 1.0, 1e-4, 37.0, 1e4])) / (np.abs(h).sum() + 1e-300)
    return (h * scale * rng.uniform(0.5, 2.0)).astype(np.float32)


def random_capture(rng, n):
    wide = rng.random() < 0.35
    full = 32767 if wide else 2047
    amp = rng.uniform(0.02, 1.0) * full
    t = np.arange(n)
    env = np.zeros(n)
    pos = 0
    while pos < n:                                  # stretches: silence, carrier, slow ramps through the threshold region
        ln = int(rng.integers(200, 20000))
        kind = rng.random()
        if kind < 0.4:
            seg = np.zeros(ln)
        elif kind < 0.7:
            seg = np.full(ln, amp)
        else:
            seg = amp * (0.5 + 0.5 * np.sin(2 * np.pi * np.arange(ln) / rng.uniform(500, 8000)))
        env[pos:pos + ln] = seg[: max(0, min(ln, n - pos))]
        pos += ln
    ph = rng.uniform(0, 2 * np.pi)
    noise = int(rng.choice([0, 3, 40, 400]))
    i = env * np.cos(ph) + (rng.integers(-noise, noise + 1, size=n) if noise else 0)
    q = env * np.sin(ph) + (rng.integers(-noise, noise + 1, size=n) if noise else 0)
    if wide and rng.random() < 0.5:
        idx = rng.integers(0, n, size=max(1, n // 200))
        i[idx] = rng.choice([-32768, 32767, 2048, -2049], size=idx.size)
    iq = np.empty(2 * n, np.int16)
    iq[0::2] = np.clip(np.round(i), -full - 1, full).astype(np.int16)
    iq[1::2] = np.clip(np.round(q), -full - 1, full).astype(np.int16)
    return iq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    tmp = tempfile.mkdtemp()
    stats = dict(cases=0, receivers=0, samples=0, recomputes=0, mfma_cases=0, valu_only_cases=0, max_err_over_scale=0.0,
                 mismatches=[])
    t_end = time.time() + args.seconds
    last = time.time()
    while time.time() < t_end:
        if time.time() - last > 30:         # (a run that stays silent for minutes is taken to be hung)
            last = time.time()
            print("[fuzz_mfma] %d cases, %d mismatches" % (stats["cases"], len(stats["mismatches"])), file=sys.stderr, flush=True)
        # one stage without decimation, or (a third of the cases) the backend default's shape: two decimate-by-2
        # stages of up to 16 and 32 taps, folded into one decimate-by-4 product on the matrix cores
        two = rng.random() < 0.33
        if two:
            t1, t2 = random_taps(rng, 16), random_taps(rng, 32)
            stages = [{"decimation": 2, "taps": [float(t) for t in t1]}, {"decimation": 2, "taps": [float(t) for t in t2]}]
            taps = np.array([np.abs(t1.astype(np.float64)).sum() * np.abs(t2.astype(np.float64)).sum()], dtype=np.float64)
            stats["two_stage_cases"] = stats.get("two_stage_cases", 0) + 1
        else:
            taps = random_taps(rng)
            stages = [{"decimation": 1, "taps": [float(t) for t in taps]}]
        n = int(rng.integers(2000, 120000 if taps.size > 64 else 400000))
        if two:
            n -= n % 4
        iq = random_capture(rng, n)
        path = os.path.join(tmp, "f.json")
        with open(path, "w") as f:
            json.dump({"filter": {"stages": stages}}, f)
        flt = ok.Filter.load(path)
        of = O.load_filter_json(path)
        spb = int(rng.choice([512, 4096, 8192]))
        y0 = O.rx(iq, of, 0.0, None, spb, want_bits=True, want_fir=True).fir.astype(np.float64)
        mag = np.sqrt(y0[:, 0] ** 2 + y0[:, 1] ** 2)
        loud = mag[mag > 0]
        if loud.size == 0:
            continue
        thr = float(np.float32(np.quantile(loud, rng.uniform(0.1, 0.9)) * rng.uniform(0.9, 1.1)))
        if not np.isfinite(thr) or thr <= 0:
            continue
        want = O.rx(iq, of, thr, None, spb, want_bits=True, want_fir=True)
        scale = float(np.abs(taps.astype(np.float64)).sum()) * float(np.abs(iq.astype(np.int32)).max()) / 2048.0
        stats["cases"] += 1
        stats["samples"] += n
        for valu in (False, True):
            for keep in (False, True):
                rx = ok.Receiver(flt, None, max_samples=n, threshold=thr, samples_per_buffer=spb, edge_capacity=n + 64,
                                 keep_fir=keep, fir_valu=valu)
                got = rx.rx(iq)
                stats["receivers"] += 1
                bits = rx.bits()
                okb = bits.size == want.bits.size and bool((bits == want.bits).all())
                okf = True
                if keep and scale > 0:
                    y = rx.fir_output().astype(np.float64)
                    err = float(np.abs(y - want.fir.astype(np.float64)).max()) / scale
                    if not valu:
                        stats["max_err_over_scale"] = max(stats["max_err_over_scale"], err)
                    okf = bool((np.abs(y - want.fir) <= 1e-5 * np.maximum(np.abs(want.fir), scale)).all())
                if not valu and not keep:
                    stats["recomputes"] += int(got.stats["guard_recomputes"])
                if not (okb and okf):
                    stats["mismatches"].append(dict(ntaps=int(taps.size), two_stage=bool(two), n=n, thr=thr, valu=valu, keep=keep, bits_ok=okb,
                                                    floats_ok=okf, seed=args.seed, case=stats["cases"],
                                                    first_diff=int(np.nonzero(bits != want.bits)[0][0]) if not okb and bits.size == want.bits.size else -1))
                rx.close()
        if len(stats["mismatches"]) > 5:
            break
    stats["seconds"] = args.seconds
    stats["seed"] = args.seed
    print(json.dumps(stats))
    sys.exit(1 if stats["mismatches"] else 0)


if __name__ == "__main__":
    main()
