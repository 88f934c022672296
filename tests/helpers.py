"""Helpers shared by the test modules (golden fixture decoding)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_path(kind: str, name: str) -> str:
    return os.path.join(GOLDEN, kind, name + ".json")


def iq_from_rle(i_rle, num_samples):
    """Golden captures store the I rail run-length encoded; Q is zero."""
    i = np.concatenate([np.full(n, v, dtype=np.int16) for v, n in i_rle])
    assert i.size == num_samples
    iq = np.zeros(2 * num_samples, dtype=np.int16)
    iq[0::2] = i
    return iq


def stream_from_runs(runs):
    out, lvl = [], 0
    for n in runs:
        out.append(np.full(n, lvl, dtype=np.uint8))
        lvl ^= 1
    return np.concatenate(out)


def edges_of(bits):
    b = np.asarray(bits).astype(np.int8)
    return np.nonzero(np.diff(np.concatenate([[0], b])))[0]


# ---- the reference FIR harness's long recipes (src/matlab/gen_samples.m:13-42) -------------------------
def harness_tone(n, period):
    """exp(1j * 2*pi/period * t), t = 1..n, as float32 I,Q pairs (gen_samples.m:19-34)"""
    import numpy as np
    t = np.arange(1, n + 1, dtype=np.float64)
    z = np.exp(1j * 2.0 * np.pi / period * t)
    return np.stack([z.real, z.imag], axis=1).astype(np.float32)


def steady_state_response(stage_taps, decims, n_in, period):
    """What a decimating FIR chain (fir.c:355-395: stage output j sits on input index D(j+1)-1) makes of
    that tone once every stage's history is full, in double precision: a tone exp(1j w (n+1)) through a
    stage with taps h and decimation D comes out as H(w) exp(1j D w (j+1)) -- the same form at D w --
    with H(w) = sum_k h[k] exp(-1j w k).  Returns (expected complex outputs, index of the first
    output whose inputs all lie inside the capture)."""
    import numpy as np
    w = 2.0 * np.pi / period
    gain = 1.0 + 0.0j
    n, first = n_in, 0
    for h, d in zip(stage_taps, decims):
        h = np.asarray(h, dtype=np.float64)
        gain *= np.sum(h * np.exp(-1j * w * np.arange(h.size)))
        # output j needs inputs D(j+1)-1-(T-1) .. D(j+1)-1 of a stream that is itself settled from `first` on
        first = -(-(first + h.size - 1 + 1) // d) - 1 + (1 if (first + h.size) % d else 0)
        first = max(first, 0)
        while d * (first + 1) - 1 - (h.size - 1) < 0:
            first += 1
        w *= d
        n //= d
    j = np.arange(n, dtype=np.float64)
    return gain * np.exp(1j * w * (j + 1.0)), first
