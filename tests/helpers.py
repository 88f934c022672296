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
