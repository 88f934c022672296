"""BASELINE.json's configurations at full size on ONE MI355X, checked through properties that do not
need the CPU oracle to finish 2^32 samples: every decoded payload was transmitted, in order, and
nearly all transmitted ones were decoded; decoding is idempotent; a capture cut into shards (halo +
carried state) and a capture pipelined in chunks (state carried on the device) decode exactly like
the whole one.  These are the runs that put 64-bit sample indices, the batched layout and the shard
protocol through their paces; each costs tens of milliseconds of GPU time.  Skipped (not failed)
only when the card has too little free memory for the case."""
import json
import os

import numpy as np
import pytest

from tests.helpers import golden_path

pytestmark = pytest.mark.gpu
RATE = 3000000


@pytest.fixture(scope="module")
def ok():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from ookiedokie_amd import build as okbuild
    okbuild.build()
    import ookiedokie_amd as okm
    okm.lib()
    return okm


def _need(gib):
    import torch
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < gib * (1 << 30):
        pytest.skip("needs %d GiB of free HBM, the card has %.0f" % (gib, free / (1 << 30)))


def _sent_in_order(syn, result):
    """every decoded payload was transmitted, in transmission order; returns how many were decoded"""
    sent = [syn.message(i)[1] for i in range(syn.num_messages)]
    j = 0
    for p in result.payloads:
        while j < len(sent) and sent[j] != bytes(p):
            j += 1
        assert j < len(sent), "decoded a payload that was never sent"
        j += 1
    # (1 message in 64 carries a glitch that makes the reference drop it; the capture may end inside one)
    assert len(result.payloads) >= 0.8 * (len(sent) - 2), (len(result.payloads), len(sent))
    return len(result.payloads)


def _capture(ok, dev, n, seed):
    import torch
    syn = ok.Synth(dev, n, seed=seed, sample_rate=RATE)
    cap = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
    syn.fill_device(cap.data_ptr())
    torch.cuda.synchronize()
    return cap, syn


def test_config2_16GiB_255_taps_unknown_remote1(ok, tmp_path):
    """configs[2]: 2^32 samples (64-bit indices everywhere), 255 real taps (hamming-windowed sinc, cutoff
    Fs/64, unity DC gain: SURVEY.md 8(d)), unknown-remote1."""
    _need(80)
    n = 1 << 32
    k = np.arange(255) - 127
    h = np.sinc(k / 32.0) * np.hamming(255)
    h = h / h.sum()
    fpath = tmp_path / "sinc255.json"
    fpath.write_text(json.dumps({"filter": {"stages": [{"decimation": 1, "taps": list(h)}]}}))
    flt = ok.Filter.load(str(fpath))
    dev = ok.Device.load(golden_path("devices", "unknown-remote1"), RATE)
    cap, syn = _capture(ok, dev, n, 0xC2)
    rx = ok.Receiver(flt, dev, max_samples=n)
    res = rx.rx_device(cap.data_ptr(), n)
    assert res.stats["decimated_samples"] == n and res.stats["fsm_path"] == 1
    decoded = _sent_in_order(syn, res)
    assert decoded > 10000
    # messages from beyond the 32-bit range, in increasing order
    assert int(res.msg_samples[-1]) > (1 << 32) - (1 << 22) and (np.diff(res.msg_samples.astype(np.int64)) > 0).all()
    again = rx.rx_device(cap.data_ptr(), n)
    assert list(again.msg_samples) == list(res.msg_samples) and (again.payloads == res.payloads).all()
    assert again.stats["num_edges"] == res.stats["num_edges"]
    rx.close()


def test_north_star_16GiB_whole_equals_pipelined(ok):
    """the bench capture (2^32 samples, fs32_fs4, p3l-nexa2012): whole, and pipelined in 1 GiB chunks"""
    _need(110)
    n = 1 << 32
    flt = ok.Filter.load(golden_path("filters", "fs32_fs4"))
    dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), RATE)
    cap, syn = _capture(ok, dev, n, 0x00C0FFEE + 2)
    rx = ok.Receiver(flt, dev, max_samples=n)
    whole = rx.rx_device(cap.data_ptr(), n)
    assert whole.stats["fsm_path"] == 1 and whole.stats["front_launches"] > 1
    _sent_in_order(syn, whole)
    rx.close()
    rxp = ok.Receiver(flt, dev, max_samples=n, pipeline_chunk_samples=1 << 28)
    piped = rxp.rx_device(cap.data_ptr(), n)
    assert piped.stats["pipeline_chunks"] >= 16 and piped.stats["fsm_path"] == 1
    assert list(piped.msg_samples) == list(whole.msg_samples) and (piped.payloads == whole.payloads).all()
    assert piped.stats["num_edges"] == whole.stats["num_edges"] and piped.stats["num_errors"] == whole.stats["num_errors"]
    rxp.close()


def test_config3_128_captures_of_64MiB_batched(ok):
    """configs[3], one GPU's share: 128 independent captures of 2^24 samples in one batched call"""
    _need(24)
    import torch
    ncap, m = 128, 1 << 24
    flt = ok.Filter.load(golden_path("filters", "fs32_fs4"))
    dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), RATE)
    stride = m + 64
    buf = torch.empty(2 * stride * ncap + 64, dtype=torch.int16, device="cuda")
    syns = []
    for c in range(ncap):
        s = ok.Synth(dev, m, seed=0xC300 + c, sample_rate=RATE)
        s.fill_device(buf.data_ptr() + 4 * stride * c)
        syns.append(s)
    torch.cuda.synchronize()
    rx = ok.Receiver(flt, dev, max_samples=m, max_captures=ncap, message_capacity=1 << 18)
    res = rx.rx_device(buf.data_ptr(), m, num_captures=ncap, stride=stride)
    assert res.stats["fsm_path"] == 1
    total = 0
    for c in (0, 1, 63, 127):
        total += _sent_in_order(syns[c], res.for_capture(c))
    assert total > 100
    # a capture decodes the same alone as in the batch
    solo = ok.Receiver(flt, dev, max_samples=m)
    for c in (5, 126):
        r1 = solo.rx_device(buf.data_ptr() + 4 * stride * c, m)
        rb = res.for_capture(c)
        assert list(r1.msg_samples) == list(rb.msg_samples) and (r1.payloads == rb.payloads).all()
    solo.close()
    rx.close()


def test_config4_8GiB_as_two_shards_equals_whole(ok):
    """configs[4] at one-GPU scale: a 2^31-sample capture as two shards (31-sample halo, 64-byte carried
    state, one refine) decodes like the whole capture"""
    _need(60)
    from ookiedokie_amd.distributed import shard_bounds
    n = 1 << 31
    flt = ok.Filter.load(golden_path("filters", "fs32_fs4"))
    dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), RATE)
    cap, syn = _capture(ok, dev, n, 0xC4)
    rx = ok.Receiver(flt, dev, max_samples=n)
    whole = rx.rx_device(cap.data_ptr(), n)
    _sent_in_order(syn, whole)
    b = shard_bounds(n, 2, 8192, flt.total_decimation)
    H = rx.halo_samples
    r0, s0 = rx.shard_begin(cap.data_ptr(), b[1], None, False, None)
    halo = cap[2 * (b[1] - H):2 * b[1]]                 # stays on the device
    rx1 = ok.Receiver(flt, dev, max_samples=n - b[1])
    r1, s1 = rx1.shard_begin(cap.data_ptr() + 4 * b[1], n - b[1], halo, True, None)
    if bytes(s0) != bytes(ok.FsmState()):               # the speculative pass assumed a reset machine
        r1, s1 = rx1.shard_refine(s0)
    got = list(r0.msg_samples) + [int(x) + b[1] // flt.total_decimation for x in r1.msg_samples]
    assert got == [int(x) for x in whole.msg_samples], "sharded != whole"
    assert (np.concatenate([r0.payloads, r1.payloads]) == whole.payloads).all()
    rx.close()
    rx1.close()


def test_config4_shard_size_32GiB_whole_equals_two_shards(ok):
    """configs[4]'s per-GPU shard size: a 2^33-sample (32 GiB) capture decoded whole -- message indices and the
    front end's tile / word arithmetic beyond 2^32 -- and as two shards of 2^32 with halo + carried state"""
    _need(110)
    from ookiedokie_amd.distributed import shard_bounds
    n = 1 << 33
    flt = ok.Filter.load(golden_path("filters", "fs32_fs4"))
    dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), RATE)
    cap, syn = _capture(ok, dev, n, 0xC5)
    rx = ok.Receiver(flt, dev, max_samples=n)
    whole = rx.rx_device(cap.data_ptr(), n)
    assert whole.stats["decimated_samples"] == n and whole.stats["fsm_path"] == 1
    decoded = _sent_in_order(syn, whole)
    assert decoded > 15000
    ms = whole.msg_samples.astype(np.int64)
    assert (np.diff(ms) > 0).all() and int(ms[-1]) > (1 << 33) - (1 << 22)
    assert int((ms > (1 << 32)).sum()) > 7000           # plenty of messages beyond the 32-bit range
    rx.close()
    b = shard_bounds(n, 2, 8192, flt.total_decimation)
    rx0 = ok.Receiver(flt, dev, max_samples=b[1])
    H = rx0.halo_samples
    r0, s0 = rx0.shard_begin(cap.data_ptr(), b[1], None, False, None)
    halo = cap[2 * (b[1] - H):2 * b[1]]
    rx1 = ok.Receiver(flt, dev, max_samples=n - b[1])
    r1, s1 = rx1.shard_begin(cap.data_ptr() + 4 * b[1], n - b[1], halo, True, None)
    if bytes(s0) != bytes(ok.FsmState()):
        r1, s1 = rx1.shard_refine(s0)
    got = list(r0.msg_samples) + [int(x) + b[1] // flt.total_decimation for x in r1.msg_samples]
    assert got == [int(x) for x in whole.msg_samples], "sharded != whole"
    assert (np.concatenate([r0.payloads, r1.payloads]) == whole.payloads).all()
    rx0.close()
    rx1.close()
