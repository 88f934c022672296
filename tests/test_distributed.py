"""Multi-process tests of the sharding layer (ookiedokie_amd/distributed.py).

CPU (gloo, world_size 2 and 3): the exchange protocol -- neighbour halo
send/recv, all-gather of carried states, refine-until-stable, message gather
-- against a stand-in engine whose shard function is known, checked against
the sequential chain.  GPU (marked): two ranks sharing cuda:0 run the real
engine over two halves of a capture and must reproduce the oracle's
single-pass result.
"""
import hashlib
import os
import socket

import numpy as np
import pytest

from tests.helpers import golden_path, iq_from_rle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _State:
    def __init__(self, raw: bytes):
        self.raw = bytes(raw).ljust(64, b"\0")[:64]

    def __bytes__(self):
        return self.raw


class FakeResult:
    def __init__(self, samples, payloads):
        self.msg_samples = np.array(samples, dtype=np.uint64)
        self.payloads = np.array(payloads, dtype=np.uint8).reshape(len(samples), 4)


class FakeEngine:
    """Shard r maps an incoming 64-byte state to an outgoing one; shards with
    `sticky` ignore their incoming state (as a shard whose state machine
    re-synchronises does).  Messages depend on the incoming state and on the
    halo, so a wrong exchange shows."""
    halo_samples = 3

    def __init__(self, rank, sticky):
        self.rank, self.sticky = rank, sticky
        self.halo = None
        self.calls = 0

    @staticmethod
    def state_from_bytes(raw):
        return _State(raw)

    @staticmethod
    def initial():
        return _State(b"init")

    def _f(self, state_in: _State):
        self.calls += 1
        h = hashlib.sha256(bytes([self.rank]) + (b"" if self.sticky else bytes(state_in))).digest()
        halo_sum = int(np.asarray(self.halo, dtype=np.int64).sum()) if self.halo is not None else -1
        msg = hashlib.sha256(bytes(state_in) + str(halo_sum).encode()).digest()
        return FakeResult([self.rank * 10 + 1], [list(msg[:4])]), _State(h)

    def shard_begin(self, ptr, n, halo, last, state_in):
        self.halo = halo
        assumed = state_in if state_in is not None else (self.initial() if self.rank == 0 else _State(b"guess"))
        return self._f(assumed)

    def shard_refine(self, state_in):
        return self._f(state_in)


def _fake_worker(rank, world, port, sticky_mask, out_q):
    import torch.distributed as dist
    from ookiedokie_amd import distributed as okd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = FakeEngine(rank, bool(sticky_mask[rank]))
    tail = np.arange(2 * 5, dtype=np.int16) + 100 * rank      # 5 samples, last 3 are the halo
    res = okd.demodulate_sharded(eng, d_iq_ptr=0, num_local_samples=1000, tail_samples=tail,
                                 decimated_offset=1000 * rank)
    allres = okd.gather_messages(res)
    if rank == 0:
        out_q.put((allres.msg_samples.tolist(), allres.payloads.tolist(), allres.rounds))
    out_q.put(("calls", rank, eng.calls))
    dist.barrier()
    dist.destroy_process_group()


def _sequential(world, sticky_mask):
    state = FakeEngine.initial()
    samples, pays = [], []
    for r in range(world):
        eng = FakeEngine(r, bool(sticky_mask[r]))
        if r > 0:
            prev_tail = np.arange(2 * 5, dtype=np.int16) + 100 * (r - 1)
            eng.halo = prev_tail[-6:]
        res, state = eng._f(state)
        samples += [int(s) + 1000 * r for s in res.msg_samples]
        pays += res.payloads.tolist()
    return samples, pays


@pytest.mark.parametrize("world,sticky", [(2, (0, 0)), (2, (1, 1)), (3, (0, 1, 0)), (3, (0, 0, 0))])
def test_sharded_protocol_matches_sequential_chain_gloo(world, sticky):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fake_worker, args=(r, world, port, sticky, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, calls = None, {}
    for _ in range(world + 1):
        item = q.get(timeout=120)
        if item[0] == "calls":
            calls[item[1]] = item[2]
        else:
            got = item
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_samples, want_pays = _sequential(world, sticky)
    assert got[0] == want_samples
    assert got[1] == want_pays
    # rank 0 never refines; nobody runs more than world+1 times
    assert calls[0] == 1 and max(calls.values()) <= world + 1


def _short_worker(rank, world, port, lens, out_q):
    """shards shorter than the halo (or empty): the halo a rank hands on is the end of
    [what it received | its own samples]"""
    import torch.distributed as dist
    from ookiedokie_amd import distributed as okd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = FakeEngine(rank, False)
    start = sum(lens[:rank])
    own = np.arange(2 * start, 2 * (start + lens[rank]), dtype=np.int16)     # sample k = (2k, 2k+1)
    okd.demodulate_sharded(eng, d_iq_ptr=0, num_local_samples=lens[rank], tail_samples=own[-2 * eng.halo_samples:],
                           decimated_offset=start)
    out_q.put((rank, None if eng.halo is None else np.asarray(eng.halo).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lens", [(5, 1, 0, 4), (2, 2, 2), (7, 0, 0)])
def test_halo_passes_through_short_and_empty_shards_gloo(lens):
    import torch.multiprocessing as mp
    world = len(lens)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_short_worker, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    halos = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H = FakeEngine.halo_samples
    stream = np.arange(2 * sum(lens), dtype=np.int16)
    assert halos[0] is None
    for r in range(1, world):
        before = np.concatenate([np.zeros(2 * H, dtype=np.int16), stream[:2 * sum(lens[:r])]])[-2 * H:]
        assert halos[r] == before.tolist(), (r, halos[r])


def test_partition_and_bounds():
    from ookiedokie_amd import distributed as okd
    # fewer alignment units than ranks: the empty shards are the trailing ones
    b = okd.shard_bounds(3 * 8192 - 5, 8, 8192, 1)
    assert b == [0, 8192, 16384, 3 * 8192 - 5, 3 * 8192 - 5, 3 * 8192 - 5, 3 * 8192 - 5, 3 * 8192 - 5, 3 * 8192 - 5]
    from ookiedokie_amd import distributed as okd
    assert okd.partition_captures(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((okd.partition_captures(1024, 8, r) for r in range(8)), [])) == list(range(1024))
    b = okd.shard_bounds(1 << 20, 8, 8192, 4)
    assert b[0] == 0 and b[-1] == 1 << 20 and all(x % 8192 == 0 for x in b[1:-1])
    assert all(b[i] < b[i + 1] for i in range(8))
    b = okd.shard_bounds(100000, 3, 1000, 4)
    assert all(x % 1000 == 0 for x in b[1:-1]) and b[-1] == 100000
    b = okd.shard_bounds(12345, 4, 1001, 4)      # lcm(1001, 4) = 4004
    assert all(x % 4004 == 0 for x in b[1:-1])


# ----------------------------------------------------------------------------- GPU

def _gpu_worker(rank, world, port, out_q, device_tail=False):
    import torch
    import torch.distributed as dist
    import ookiedokie_amd as ok
    from ookiedokie_amd import distributed as okd
    import json
    from tests.helpers import GOLDEN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        g = json.load(f)["G1"]
    iq = iq_from_rle(g["i_rle"], g["num_samples"])
    rng = np.random.default_rng(21)
    iq = (iq + rng.integers(-40, 41, size=iq.size)).astype(np.int16)
    n = iq.size // 2
    flt = ok.Filter.load(golden_path("filters", "fs128_fs16_dec4"))
    dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), 3000000 // 4)
    bounds = okd.shard_bounds(n, world, 8192, flt.total_decimation)
    lo, hi = bounds[rank], bounds[rank + 1]
    t = torch.from_numpy(iq[2 * lo:2 * hi].copy()).cuda()
    rx = ok.Receiver(flt, dev, max_samples=hi - lo, samples_per_buffer=8192)
    if device_tail:
        # the nccl / RCCL shape of the call on one GPU: tail and received halo are DEVICE tensors (gloo moves
        # them through the host underneath, but shard_begin gets a device pointer it copies on its own stream)
        H = int(rx.halo_samples)
        tail = t[2 * (hi - lo - H):] if H else t[:0]
        res = okd.demodulate_sharded(rx, d_iq_ptr=t.data_ptr(), num_local_samples=hi - lo, tail_samples=tail,
                                     decimated_offset=lo // flt.total_decimation,
                                     comm_device=torch.device("cuda", 0))
    else:
        res = okd.demodulate_sharded(rx, d_iq_ptr=t.data_ptr(), num_local_samples=hi - lo,
                                     tail_samples=iq[2 * lo:2 * hi], decimated_offset=lo // flt.total_decimation)
    allres = okd.gather_messages(res)
    if rank == 0:
        out_q.put((allres.msg_samples.tolist(), [bytes(p).hex() for p in allres.payloads]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("device_tail", [False, True])
def test_two_ranks_one_capture_matches_oracle(oracle, vectors, device_tail):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q, device_tail)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = vectors["G1"]
    iq = iq_from_rle(g["i_rle"], g["num_samples"])
    rng = np.random.default_rng(21)
    iq = (iq + rng.integers(-40, 41, size=iq.size)).astype(np.int16)
    of = oracle.load_filter_json(golden_path("filters", "fs128_fs16_dec4"))
    od = oracle.load_device_json(golden_path("devices", "p3l-nexa2012"), 750000)[0]
    want = oracle.rx(iq, of, 0.1, od, 8192)
    assert got[0] == [int(s) for s in want.msg_samples] and len(got[0]) == 3
    assert got[1] == [bytes(p).hex() for p in want.payloads]


# ------------------------------------------------- bench.py --gpus 2 as fresh child processes

def _bench_line(args, timeout=600):
    """python bench.py <args> in a fresh process (nothing in it has touched a GPU when it spawns its ranks)"""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_sharded_on_one_gpu():
    """`bench.py --gpus 2 --workload sharded`: two ranks (gloo: they share this box's GPU) cut ONE capture in two,
    exchange halo + carried state, and the line's own check says the sharded result equals the whole capture's."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("rehearsal of the two-rank path on ONE device")
    d = _bench_line(["--gpus", "2", "--workload", "sharded", "--backend", "gloo", "--samples", "16777216",
                     "--steps", "2", "--warmup", "1", "--allow-shared-gpu"])
    assert d["n_gpus"] == 2 and d["config"]["backend"] == "gloo" and d["config"]["shards"] == 2
    assert d["config"]["samples_per_shard"] == 16777216
    assert d["config"]["sharded_equals_whole"] is True and d["config"]["messages"] > 0
    assert d["scaling"] == "weak" and d["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_batch_on_one_gpu():
    """`bench.py --gpus 2 --workload batch`: independent captures per rank, no data-path collective"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("rehearsal of the two-rank path on ONE device")
    d = _bench_line(["--gpus", "2", "--workload", "batch", "--backend", "gloo", "--samples", "1048576",
                     "--steps", "2", "--warmup", "1", "--contexts", "1", "--no-sub-records", "--allow-shared-gpu"])
    assert d["n_gpus"] == 2 and d["config"]["backend"] == "gloo"
    assert d["config"]["captures_per_step"] == 128 and d["config"]["messages_per_step"] >= 0
    assert d["value"] > 0
