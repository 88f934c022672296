#!/usr/bin/env python3
"""Regenerates tests/golden/vectors.json.  Run in the build container only
(it needs oracle/_ref, i.e. /root/reference):

    python tests/golden/make_golden.py

What comes from where
---------------------
* Waveforms G1/G2/G3 are produced by the REFERENCE's own tx code compiled
  into oracle/_ref (sm_generate, state_machine.c:825-873, and
  complexf_to_sc16q11, complexf.h:87-96), laid out exactly as ookiedokie_tx
  does (ookiedokie.c:301-344: tx_delay zeros then the message, per repeat).
  They are stored run-length encoded (the reference tx emits I in {0,1945},
  Q = 0).
* "survey" expectations are the numbers SURVEY.md section 8(c) recorded from
  the real ookiedokie binary (rx through fs32_fs4, threshold 0.1, 8192
  samples per buffer): edge counts / first edges / OUTPUT_READY indices /
  payloads.  This script asserts the oracle reproduces every one of them
  before writing the file.
* "ref_fsm" expectations are produced here by the reference's real
  state_machine.c (oracle/_ref) on raw 0/1 streams ("-F none" shape):
  G7 tolerance boundaries and a set of seeded random streams with
  glitches, for several buffer lengths and sample rates.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle as O  # noqa: E402

RATE = 3000000
G1_BITS = "100111110101010100001101011100000000"   # Channel=2, 21.5 C
G2_BITS = "01011101010000100110000010011111"       # Button=P2, ID=0x42


def bits_to_bytes(s):
    a = np.array([int(c) for c in s], dtype=np.uint8)
    return np.packbits(a, bitorder="little").tobytes()


def rle(level):
    level = np.asarray(level, dtype=np.int64)
    change = np.nonzero(np.diff(level))[0] + 1
    starts = np.concatenate([[0], change])
    lens = np.diff(np.concatenate([starts, [level.size]]))
    return [[int(level[s]), int(n)] for s, n in zip(starts, lens)]


def tx_capture(dev_json, bits, repeats, delay=12000):
    dev, _ = O.load_device_json(dev_json, RATE)
    wave = O.RefSm(dev).generate(bits_to_bytes(bits), 0.95)
    rep = np.concatenate([np.zeros((delay, 2), np.float32), wave])
    iq = O.ref_pack(np.concatenate([rep] * repeats))
    assert not iq[1::2].any()
    return dev, iq


def edges_of(bits):
    b = bits.astype(np.int8)
    return np.nonzero(np.diff(np.concatenate([[0], b])))[0]


def run_lengths_stream(runs):
    """runs: list of lengths, level alternates starting at 0."""
    out = []
    lvl = 0
    for n in runs:
        out.append(np.full(n, lvl, dtype=np.uint8))
        lvl ^= 1
    return np.concatenate(out)


def p3l_runs(bits, start_pulse=1500, first_gap=26100, pulse=1500,
             gap0=6000, gap1=12000, lead=12000):
    runs = [lead, start_pulse, first_gap]
    for c in bits:
        runs += [pulse, gap1 if c == "1" else gap0]
    runs += [pulse, 4000]
    return runs


def main():
    O.build()
    assert O.have_ref(), "needs oracle/_ref (reference tree)"
    fir = O.load_filter_json(os.path.join(HERE, "filters", "fs32_fs4.json"))
    fir_dec4 = O.load_filter_json(os.path.join(HERE, "filters", "fs128_fs16_dec4.json"))
    p3l = os.path.join(HERE, "devices", "p3l-nexa2012.json")
    rem = os.path.join(HERE, "devices", "unknown-remote1.json")
    out = {"rate": RATE, "threshold": 0.1}

    # ---- G1 ---------------------------------------------------------------
    dev1, iq1 = tx_capture(p3l, G1_BITS, 3)
    assert iq1.size // 2 == 1221300
    r = O.rx(iq1, fir, 0.1, dev1, 8192, want_bits=True)
    e = edges_of(r.bits)
    assert len(e) == 228 and list(e[:6]) == [12014, 13517, 39614, 41117, 53114, 54617]
    assert list(r.msg_samples) == [405615, 812715, 1219815]
    assert all(r.payload_bits(i, 36) == G1_BITS for i in range(3))
    for spb in (1000, 4096, 65536):
        assert list(O.rx(iq1, fir, 0.1, dev1, spb).msg_samples) == [405615, 812715, 1219815]
    r4 = O.rx(iq1, fir_dec4, 0.1, dev1.with_rate(RATE // 4), 8192)
    assert len(r4.msg_samples) == 3     # survey: same 3 messages with dec-4
    out["G1"] = {
        "device": "p3l-nexa2012", "filter": "fs32_fs4", "spb": 8192,
        "i_rle": rle(iq1[0::2]), "num_samples": 1221300,
        "survey": {"num_edges": 228,
                   "first_edges": [12014, 13517, 39614, 41117, 53114, 54617],
                   "msg_samples": [405615, 812715, 1219815],
                   "payload_bits": G1_BITS,
                   "same_msgs_at_spb": [1000, 4096, 65536]},
        "oracle": {"edges": [int(v) for v in e],
                   "dec4_msg_samples": [int(v) for v in r4.msg_samples]},
    }

    # ---- G2 ---------------------------------------------------------------
    dev2, iq2 = tx_capture(rem, G2_BITS, 2)
    assert iq2.size * 2 == 1687200
    r = O.rx(iq2, fir, 0.1, dev2, 8192, want_bits=True)
    assert list(r.msg_samples) == [209265, 420165]
    assert all(r.payload_bits(i, 32) == G2_BITS for i in range(2))
    out["G2"] = {
        "device": "unknown-remote1", "filter": "fs32_fs4", "spb": 8192,
        "i_rle": rle(iq2[0::2]), "num_samples": iq2.size // 2,
        "survey": {"msg_samples": [209265, 420165], "payload_bits": G2_BITS},
        "oracle": {"edges": [int(v) for v in edges_of(r.bits)]},
    }

    # ---- G3: glitch => FSM error => rest of buffer dropped -----------------
    iq3 = iq1.copy()
    iq3[2 * 9000:2 * 9300:2] = 1945
    g3 = {}
    for spb, want in ((1000, 3), (4096, 0), (8192, 0), (65536, 0)):
        rr = O.rx(iq3, fir, 0.1, dev1, spb)
        assert len(rr.msg_samples) == want, (spb, rr.msg_samples)
        g3[str(spb)] = {"num_msgs": want,
                        "msg_samples": [int(v) for v in rr.msg_samples],
                        "err_samples": [int(v) for v in rr.err_samples]}
    out["G3"] = {"base": "G1", "glitch": [9000, 9300, 1945],
                 "survey_num_msgs": {"1000": 3, "4096": 0, "8192": 0, "65536": 0},
                 "oracle": g3}

    # ---- G7: tolerance boundaries on the REAL reference FSM ----------------
    # SURVEY.md 8(c) G7 (real binary, -F none): run length L is tested with
    # k = L-1.  Expectations below are the survey's; we assert the compiled
    # reference FSM and the oracle FSM both give them.
    g7 = []
    cases = [("start_pulse", 1276, 0), ("start_pulse", 1277, 1),
             ("start_pulse", 1726, 1), ("start_pulse", 1727, 0),
             ("first_gap", 22186, 0), ("first_gap", 22187, 1),
             ("first_gap", 30015, 1), ("first_gap", 30016, 0),
             ("gap0", 5101, 0), ("gap0", 5102, 1),
             ("gap0", 6901, 1), ("gap0", 6902, 0),
             ("gap1", 10200, 0), ("gap1", 10201, 1),
             ("gap1", 13800, 1), ("gap1", 13801, 0)]
    for key, val, want in cases:
        runs = p3l_runs(G1_BITS, **{key: val})
        stream = run_lengths_stream(runs)
        ms, pay, es = O.RefSm(dev1).stream(stream, 8192)
        oms, opay, oes = O.sm_stream(dev1, stream, 8192)
        assert len(ms) == want, (key, val, len(ms))
        assert list(ms) == list(oms) and list(es) == list(oes)
        g7.append({"param": key, "value": val, "num_msgs": want,
                   "msg_samples": [int(v) for v in ms]})
    out["G7"] = {"device": "p3l-nexa2012", "payload_bits": G1_BITS,
                 "spb": 8192, "cases": g7}

    # ---- random glitchy streams through the REAL reference FSM -------------
    rnd = []
    for seed in range(12):
        rng = np.random.default_rng(1000 + seed)
        devp = p3l if seed % 2 == 0 else rem
        rate = [3000000, 2000000, 1000000, 750000][seed % 4]
        dev, _ = O.load_device_json(devp, rate)
        nbits = dev.max_bits
        scale = rate / 3e6
        runs = []
        for _m in range(6):
            bits = "".join(str(int(b)) for b in rng.integers(0, 2, nbits))
            if seed % 2 == 0:
                rr = p3l_runs(bits, lead=int(rng.integers(3000, 40000)))
            else:
                rr = [int(rng.integers(3000, 40000)), 26700, 13200]
                for c in bits:
                    rr += [1650, 5100 if c == "1" else 1650]
                rr += [1650, 3000]
            rr = [max(1, int(round(v * scale * rng.uniform(0.9, 1.1)))) for v in rr]
            # sprinkle glitches: split a run by a short opposite-level pulse
            for _g in range(int(rng.integers(0, 3))):
                j = int(rng.integers(0, len(rr)))
                if rr[j] > 40:
                    a = int(rng.integers(1, rr[j] - 20))
                    g = int(rng.integers(1, 15))
                    rr[j:j + 1] = [a, g, rr[j] - a - g]
            if len(rr) % 2:
                rr.append(int(rng.integers(100, 3000)))
            runs += rr
        stream = run_lengths_stream(runs)
        per = {}
        for buf in (512, 1000, 8192, 65536):
            ms, pay, es = O.RefSm(dev).stream(stream, buf)
            oms, opay, oes = O.sm_stream(dev, stream, buf)
            assert list(ms) == list(oms) and list(es) == list(oes), (seed, buf)
            assert (pay == opay).all()
            per[str(buf)] = {"msg_samples": [int(v) for v in ms],
                             "payloads": [bytes(p).hex() for p in pay],
                             "err_samples": [int(v) for v in es]}
        rnd.append({"device": os.path.basename(devp)[:-5], "rate": rate,
                    "runs": runs, "ref_fsm": per})
    out["random_streams"] = rnd

    # ---- G5: integer sample-count windows -----------
    out["G5"] = {
        "rate": RATE,
        "survey_windows": {"500": [1276, 1725], "550": [1403, 1897],
                           "1700": [4336, 5865], "2000": [5101, 6900],
                           "4000": [10200, 13799], "4400": [11220, 15180],
                           "8700": [22186, 30014], "8900": [22696, 30704]},
        "survey_timeouts": {"1000": 3000, "1100": 3300, "1500": 4501,
                            "3400": 10200, "6000": 18001, "8800": 26401,
                            "16400": 49200, "17800": 53400},
    }
    for d, (lo, hi) in out["G5"]["survey_windows"].items():
        assert O.duration_window(RATE, int(d)) == (lo, hi), d
    for t, k in out["G5"]["survey_timeouts"].items():
        assert O.timeout_count(RATE, int(t)) == k, t

    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote vectors.json", os.path.getsize(os.path.join(HERE, "vectors.json")), "bytes")


if __name__ == "__main__":
    main()
