#!/usr/bin/env python3
"""BASELINE.json configs[2], [3], [4] at the scale ONE MI355X allows (the
bench line is configs[1]; these are parity / capacity cases, timed for
DESIGN.md).  Each case checks size-independent properties: every decoded
payload was transmitted, in order; decode is idempotent; sharded == whole.

    python tools/run_configs.py [--quick] > gpurun_out/configs.json

  c2  16 GiB capture (2^32 samples), 255-tap FIR, unknown-remote1
  c3  one GPU's share of "1024 x 64 MiB captures over 8 GPUs": 128 captures
      of 2^24 samples in one batched call
  c4  one GPU's share of "256 GiB over 8 GPUs" is a 32 GiB shard: here a
      8 GiB capture is demodulated whole and as 2 shards with halo + carried
      state (the same calls distributed.py makes per rank)
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RATE = 3_000_000


def golden(kind, name):
    return os.path.join(ROOT, "tests", "golden", kind, name + ".json")


def check_sent_in_order(syns, result, caps=None):
    """every decoded payload of capture c was transmitted in capture c, in order"""
    import numpy as np
    decoded = 0
    for c, syn in enumerate(syns):
        sent = [syn.message(i)[1] for i in range(syn.num_messages)]
        r = result.for_capture(c) if caps else result
        j = 0
        for p in r.payloads:
            while j < len(sent) and sent[j] != bytes(p):
                j += 1
            assert j < len(sent), "decoded a payload that was never sent (capture %d)" % c
            j += 1
        decoded += len(r.payloads)
        assert len(r.payloads) >= 0.8 * (len(sent) - 2), (c, len(r.payloads), len(sent))
    return decoded


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="1/16 of the sizes")
    args = ap.parse_args()
    import numpy as np
    import torch
    import ookiedokie_amd as ok
    scale = 16 if args.quick else 1
    out = {}

    # ---- c2: 16 GiB, 255 taps, unknown-remote1 -------------------------------------------
    n = (1 << 32) // scale
    k = np.arange(255) - 127
    h = np.sinc(k / 32.0) * np.hamming(255)
    h = h / h.sum()
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"filter": {"stages": [{"decimation": 1, "taps": list(h)}]}}, f)
        fpath = f.name
    flt = ok.Filter.load(fpath)
    os.unlink(fpath)
    dev = ok.Device.load(golden("devices", "unknown-remote1"), RATE)
    syn = ok.Synth(dev, n, seed=0xC2, sample_rate=RATE)
    cap = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
    syn.fill_device(cap.data_ptr())
    torch.cuda.synchronize()
    rx = ok.Receiver(flt, dev, max_samples=n, threshold=0.1, samples_per_buffer=8192)
    res = rx.rx_device(cap.data_ptr(), n)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        rx.process_device(cap.data_ptr(), n)
        t.append(time.perf_counter() - t0)
    again = rx.result()
    assert list(res.msg_samples) == list(again.msg_samples) and (res.payloads == again.payloads).all()
    dec = check_sent_in_order([syn], res)
    st = again.stats
    out["c2"] = {"samples": n, "taps": 255, "device": "unknown-remote1", "seconds": round(min(t), 5),
                 "Msamples_per_s": round(n / min(t) / 1e6, 1), "fir_kernel_ms": round(st["fir_kernel_ms"], 3),
                 "messages": dec, "edges": int(st["num_edges"]), "fsm_path": int(st["fsm_path"])}
    rx.close()
    del cap
    torch.cuda.empty_cache()

    # ---- c3: 128 captures x 64 MiB, batched ----------------------------------------------------
    ncap = 128 // scale
    m = 1 << 24
    flt = ok.Filter.load(golden("filters", "fs32_fs4"))
    dev = ok.Device.load(golden("devices", "p3l-nexa2012"), RATE)
    stride = m + 64
    buf = torch.empty(2 * stride * ncap + 64, dtype=torch.int16, device="cuda")
    syns = []
    for c in range(ncap):
        s = ok.Synth(dev, m, seed=0xC300 + c, sample_rate=RATE)
        s.fill_device(buf.data_ptr() + 4 * stride * c)
        syns.append(s)
    torch.cuda.synchronize()
    rx = ok.Receiver(flt, dev, max_samples=m, max_captures=ncap, threshold=0.1, samples_per_buffer=8192,
                     message_capacity=1 << 18)
    res = rx.rx_device(buf.data_ptr(), m, num_captures=ncap, stride=stride)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        rx.process_device(buf.data_ptr(), m, num_captures=ncap, stride=stride)
        t.append(time.perf_counter() - t0)
    dec = check_sent_in_order(syns, res, caps=True)
    out["c3"] = {"captures": ncap, "samples_per_capture": m, "seconds": round(min(t), 5),
                 "Msamples_per_s": round(ncap * m / min(t) / 1e6, 1), "messages": dec,
                 "fsm_path": int(rx.stats()["fsm_path"])}
    rx.close()
    del buf
    torch.cuda.empty_cache()

    # ---- c4: one capture as 2 shards with halo + carried state == whole ------------------------------
    n = (1 << 31) // scale
    syn = ok.Synth(dev, n, seed=0xC4, sample_rate=RATE)
    cap = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
    syn.fill_device(cap.data_ptr())
    torch.cuda.synchronize()
    rx = ok.Receiver(flt, dev, max_samples=n, threshold=0.1, samples_per_buffer=8192)
    whole = rx.rx_device(cap.data_ptr(), n)
    check_sent_in_order([syn], whole)
    from ookiedokie_amd.distributed import shard_bounds
    b = shard_bounds(n, 2, 8192, flt.total_decimation)
    H = rx.halo_samples
    t0 = time.perf_counter()
    r0, s0 = rx.shard_begin(cap.data_ptr(), b[1], None, False, None)
    halo = cap[2 * (b[1] - H):2 * b[1]].cpu().numpy()
    rx1 = ok.Receiver(flt, dev, max_samples=n - b[1], threshold=0.1, samples_per_buffer=8192)
    r1, s1 = rx1.shard_begin(cap.data_ptr() + 4 * b[1], n - b[1], halo, True, None)
    rounds = 0
    if bytes(s0) != bytes(ok.FsmState()):       # speculative pass assumed a reset machine
        r1, s1 = rx1.shard_refine(s0)
        rounds = 1
    shard_s = time.perf_counter() - t0
    got_samples = list(r0.msg_samples) + [int(x) + b[1] // flt.total_decimation for x in r1.msg_samples]
    assert got_samples == [int(x) for x in whole.msg_samples], "sharded != whole"
    assert (np.concatenate([r0.payloads, r1.payloads]) == whole.payloads).all()
    out["c4"] = {"samples": n, "shards": 2, "halo_samples": int(H), "state_bytes": 64, "refine_rounds": rounds,
                 "seconds_both_shards_on_one_gpu": round(shard_s, 5), "messages": len(got_samples)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
