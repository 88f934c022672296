#!/usr/bin/env python3
"""Randomised differential run on the GPU box: HIP path (scan with span tables, scan with
per-span simulation, rounds) against the CPU oracle on message-like streams whose run
lengths sit on and around the devices' tolerance windows, with glitches, for several sample
rates and buffer sizes.  Not part of the test suite (minutes); prints a JSON summary.

    python tools/fuzz_gpu.py [--seconds 300] [--seed 1] [--sync-walk]

--sync-walk: the scan's default legs find the entry states by the walk from synchronising spans whatever the edge
count (by default only edge lists of 20 000 and more do), and one more leg runs the composing kernels
(OOKD_RX_SCAN_TABLES); the summary counts which form each run ended in (scan_entry_form).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ookiedokie_amd as ok          # noqa: E402
import oracle as O                   # noqa: E402  (checker)
from tests.helpers import golden_path, stream_from_runs   # noqa: E402

# canonical run lengths in microseconds: lead, start pulse, first gap, pulse, gap0, gap1
SHAPES = {
    "p3l-nexa2012": dict(bits=36, start=500, first=8700, pulse=500, gap0=2000, gap1=4000),
    "unknown-remote1": dict(bits=32, start=8900, first=4400, pulse=550, gap0=550, gap1=1700),
}


def jitter(rng, us, rate, mode):
    n = us * rate / 1e6
    if mode == 0:
        f = 1.0
    elif mode == 1:                              # around the +-15 % window edges
        f = rng.choice([0.85, 1.15]) + rng.uniform(-0.004, 0.004)
    elif mode == 2:
        f = rng.uniform(0.8, 1.2)
    else:
        f = rng.uniform(0.3, 3.0)
    return max(1, int(round(n * f)))


def glitched(rng, runs, rate):
    """Short pulses inside low runs (every other run, starting with the second): the edges of
    such a pulse usually fire nothing -- the state machine's counter runs on."""
    out = []
    for i, r in enumerate(runs):
        if i % 2 == 1 and r > 40 and rng.random() < 0.15:
            a = int(rng.integers(1, r - 2))
            g = int(rng.integers(1, min(r - a, max(2, int(400 * rate / 3e6)))))
            out += [a, g, max(1, r - a - g)]
        else:
            out.append(r)
    return out


def message_runs(rng, shape, rate):
    p_bad = rng.choice([0.0, 0.02, 0.1])
    def j(us):
        r = rng.random()
        mode = 0 if r >= p_bad else int(rng.integers(1, 4))
        return jitter(rng, us, rate, mode)
    runs = [j(shape["start"]), j(shape["first"])]
    nbits = shape["bits"] if rng.random() < 0.8 else int(rng.integers(1, shape["bits"] + 8))
    for _ in range(nbits):
        runs += [j(shape["pulse"]), j(shape["gap1"] if rng.random() < 0.5 else shape["gap0"])]
    runs += [j(shape["pulse"])]
    return runs


def capture_runs(rng, shape, rate):
    runs = [int(rng.integers(1, 20000))]         # leading off time
    many = rate <= 1000000 and rng.random() < 0.3      # now and then a long capture: > 1024 edges
    for _ in range(int(rng.integers(20, 45)) if many else int(rng.integers(1, 7))):
        if rng.random() < 0.3:                   # glitch pulse before the message
            runs += [max(1, int(rng.integers(1, 400) * rate / 3e6)), max(1, int(rng.integers(100, 9000) * rate / 3e6))]
        msg = message_runs(rng, shape, rate)
        runs += glitched(rng, msg, rate) if rng.random() < 0.5 else msg
        gap_us = rng.choice([300, 1000, 4000, 12000, 20000]) * rng.uniform(0.8, 1.3)
        runs += [max(1, int(gap_us * rate / 1e6))]
    return runs


def iq_from_stream(stream, rng, noise):
    iq = np.zeros(2 * stream.size, dtype=np.int16)
    iq[0::2] = stream.astype(np.int16) * 1945
    if noise:
        iq = (iq + rng.integers(-40, 41, size=iq.size)).astype(np.int16)
    return iq


def random_device_case(rng, stats):
    """A random (mostly nonsensical) state machine -- every trigger kind, reset without
    'always', zero-duration windows -- over random run lengths around its time constants."""
    from tests.test_oracle import _random_fsm
    rate = int(rng.choice([3000000, 1000000, 750000, 48000]))
    od = _random_fsm(O, rng, rate)
    d = ok.Device.from_tables(
        max_bits=od.max_bits, sample_rate=rate, state_duration_us=od.state_duration_us,
        state_timeout_us=od.state_timeout_us, trig_begin=od.trig_begin, trig_cond=od.trig_cond,
        trig_action=od.trig_action, trig_next=od.trig_next, trig_duration_us=od.trig_duration_us)
    scale = rate / 1e6
    nruns = int(rng.integers(20, 2500 if rng.random() < 0.2 else 300))
    runs = [max(1, int(rng.choice([30, 60, 100, 200, 250, 400, 1000, 5000]) * scale * rng.uniform(0.8, 1.2)))
            for _ in range(nruns)]
    stream = stream_from_runs(runs)
    if stream.size > 1_000_000:
        return
    iq = iq_from_stream(stream, rng, noise=False)
    spb = int(rng.choice([97, 512, 4096, 8192]))
    want = O.rx(iq, None, 0.1, od, spb, msg_cap=1 << 20)
    stats["cases"] += 1
    stats["random_device_cases"] = stats.get("random_device_cases", 0) + 1
    stats["messages"] += len(want.msg_samples)
    stats["errors"] += len(want.err_samples)
    for fsm_rounds, scan_sims in ((False, False), (False, True), (True, False)):
        rx = ok.Receiver(None, d, max_samples=iq.size // 2, samples_per_buffer=spb, segment_buffers=2,
                         message_slots=2 * spb * 2 + 2, message_capacity=1 << 20, edge_capacity=iq.size,
                         fsm_rounds=fsm_rounds, scan_sims=scan_sims)
        try:
            got = rx.rx(iq)
        except ok.OokdError as e:                   # a device that emits a message on (nearly) every sample
            if "overflow" not in str(e):
                raise
            stats["overflows"] = stats.get("overflows", 0) + 1
            rx.close()
            continue
        stats["receivers"] += 1
        if not fsm_rounds and not scan_sims and got.stats["fsm_path"] == 1:
            key = "random_device_entry_form_%d" % got.stats["scan_entry_form"]
            stats[key] = stats.get(key, 0) + 1
        if not fsm_rounds:
            stats["scan_runs"] += 1
            if got.stats["fsm_path"] != 1:
                stats["scan_refused"] += 1
                key = "random device reason %#x%s" % (got.stats["fsm_fallback_reason"], " sims" if scan_sims else "")
                stats["refusals"][key] = stats["refusals"].get(key, 0) + 1
        good = (list(got.msg_samples) == list(want.msg_samples) and (got.payloads == want.payloads).all()
                and got.stats["num_errors"] == len(want.err_samples))
        if not good:
            stats["mismatches"].append(dict(device="random", rate=rate, spb=spb, runs=[int(r) for r in runs[:400]],
                                            fsm_rounds=fsm_rounds, scan_sims=scan_sims,
                                            tables=dict(dur=[int(x) for x in od.state_duration_us],
                                                        to=[int(x) for x in od.state_timeout_us],
                                                        tb=[int(x) for x in od.trig_begin],
                                                        cond=[int(x) for x in od.trig_cond],
                                                        act=[int(x) for x in od.trig_action],
                                                        nxt=[int(x) for x in od.trig_next],
                                                        tdur=[int(x) for x in od.trig_duration_us],
                                                        max_bits=int(od.max_bits)),
                                            want=[int(x) for x in want.msg_samples[:20]],
                                            got=[int(x) for x in got.msg_samples[:20]],
                                            want_errors=len(want.err_samples), got_errors=int(got.stats["num_errors"])))
        rx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--random-devices", type=float, default=0.0, help="share of cases run on random state machines")
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many captures (0 = run for --seconds)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sync-walk", action="store_true")
    args = ap.parse_args()
    if args.sync_walk:
        os.environ["OOKD_DEVELOPER"] = "1"
        os.environ["OOKD_SYNC_MIN_EDGES"] = "0"
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    stats = dict(cases=0, receivers=0, messages=0, errors=0, scan_runs=0, scan_refused=0, refusals={}, mismatches=[])
    flts = {None: (None, None)}
    for name in ("fs32_fs4", "fs128_fs16_dec4"):
        flts[name] = (ok.Filter.load(golden_path("filters", name)), O.load_filter_json(golden_path("filters", name)))
    last = time.time()
    while (stats["cases"] < args.cases if args.cases else time.time() < t_end) and len(stats["mismatches"]) < 5:
        if args.random_devices and rng.random() < args.random_devices:
            random_device_case(rng, stats)
            continue
        name = str(rng.choice(list(SHAPES)))
        rate = int(rng.choice([3000000, 2000000, 1000000, 750000, 250000]))
        fname = rng.choice([None, None, "fs32_fs4", "fs128_fs16_dec4"])
        fname = None if fname is None else str(fname)
        f, of = flts[fname]
        dec = of.total_decimation if of else 1
        runs = capture_runs(rng, SHAPES[name], rate)
        stream = stream_from_runs(runs)
        if stream.size > 3_000_000:
            continue
        iq = iq_from_stream(stream, rng, noise=fname is not None)
        d = ok.Device.load(golden_path("devices", name), rate // dec)
        od, _ = O.load_device_json(golden_path("devices", name), rate // dec)
        spb = int(rng.choice([97, 1000, 4096, 8192, 65536]))
        if dec > 1:
            spb = max(dec, spb - spb % dec)
        want = O.rx(iq, of, 0.1, od, spb, want_bits=True)
        stats["cases"] += 1
        stats["messages"] += len(want.msg_samples)
        stats["errors"] += len(want.err_samples)
        # scan with span tables / with per-span simulation, rounds, and the scan pipelined in chunks of a few
        # buffers (state carried on the device)
        legs = [(False, False, 0, False), (False, True, 0, False), (True, False, 0, False),
                (False, False, int(rng.choice([2, 4, 16])) * spb, False)]
        if args.sync_walk:
            legs.append((False, False, 0, True))
        for fsm_rounds, scan_sims, chunk, tables in legs:
            rx = ok.Receiver(f, d, max_samples=iq.size // 2, samples_per_buffer=spb, fsm_rounds=fsm_rounds,
                             quiet_skip=not fsm_rounds, scan_sims=scan_sims, pipeline_chunk_samples=chunk, scan_tables=tables)
            got = rx.rx(iq)
            stats["receivers"] += 1
            if not fsm_rounds and not scan_sims and not tables and got.stats["fsm_path"] == 1:
                key = "entry_form_%d%s" % (got.stats["scan_entry_form"], "_pipelined" if got.stats["pipeline_chunks"] else "")
                stats[key] = stats.get(key, 0) + 1
            if chunk:
                stats["pipelined"] = stats.get("pipelined", 0) + (1 if got.stats["pipeline_chunks"] else 0)
                stats["pipelined_refused"] = stats.get("pipelined_refused", 0) + (
                    1 if (not got.stats["pipeline_chunks"] and got.stats["fsm_fallback_reason"]) else 0)
            elif not fsm_rounds:
                stats["scan_runs"] += 1
                if got.stats["fsm_path"] != 1:
                    stats["scan_refused"] += 1
                    key = "reason %#x spb %d%s" % (got.stats["fsm_fallback_reason"], spb, " sims" if scan_sims else "")
                    stats["refusals"][key] = stats["refusals"].get(key, 0) + 1
            errs, nerr = rx.errors()
            good = ((rx.bits() == want.bits).all() and list(got.msg_samples) == list(want.msg_samples)
                    and (got.payloads == want.payloads).all() and nerr == len(want.err_samples)
                    and (nerr > 32 or list(errs) == list(want.err_samples)))
            if not good:
                stats["mismatches"].append(dict(device=name, rate=rate, filter=fname, spb=spb, runs=[int(r) for r in runs],
                                                fsm_rounds=fsm_rounds, scan_sims=scan_sims, chunk=chunk, tables=tables,
                                                entry_form=int(got.stats["scan_entry_form"]),
                                                want=[int(x) for x in want.msg_samples],
                                                got=[int(x) for x in got.msg_samples]))
            rx.close()
        if time.time() - last > 30:
            last = time.time()
            print("[fuzz] %d cases, %d messages, %d fsm errors, %d mismatches" %
                  (stats["cases"], stats["messages"], stats["errors"], len(stats["mismatches"])), file=sys.stderr, flush=True)
    print(json.dumps(stats))
    return 1 if stats["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
