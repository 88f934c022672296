"""Multi-GPU layout of the rx path: one process per GPU, ``torch.distributed``
(backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in CPU tests).

Two ways the work shards:

* **Independent captures** (BASELINE config 4): a capture is an independent
  unit (own zero FIR history, own reset state machine), so captures are dealt
  round-robin to ranks and nothing is exchanged on the data path.
  :func:`partition_captures`.

* **One oversized capture** (BASELINE config 5): contiguous shards, cut at
  multiples of ``lcm(samples_per_buffer, total_decimation)`` so the decimation
  phase and the drop-rest-of-buffer grid (src/device.c:646) are shard-local.
  There is exactly one exchange step:

  1. *halo*: rank r sends its last ``halo_samples`` input samples to rank
     r+1 (what the FIR history, src/fir.c:49-54, would have carried) --
     ~124 B for fs32_fs4, point-to-point, latency bound;
  2. every rank demodulates its shard assuming the state machine enters it
     in its quiet state;
  3. the 64-byte carried state (what ``struct state_machine`` holds across
     ``sm_process`` calls, src/state_machine.c:57-75) of every shard is
     all-gathered; a rank whose true incoming state differs from what it
     assumed refines its shard; repeat until no rank changes (in practice
     one or two rounds: a shard's outgoing state rarely depends on its
     incoming one).  No all-reduce of data, no bandwidth-bound collective.

The engine (``ookiedokie_amd.Receiver``) is passed in, so the protocol itself
is testable on CPU with gloo and a stand-in engine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

STATE_BYTES = 64


def partition_captures(num_captures: int, world_size: int, rank: int) -> List[int]:
    """Capture i -> rank i mod world_size."""
    return list(range(rank, num_captures, world_size))


def shard_bounds(num_samples: int, world_size: int, samples_per_buffer: int,
                 total_decimation: int) -> List[int]:
    """Contiguous shard boundaries, every inner one a multiple of
    lcm(samples_per_buffer, total_decimation).  The alignment units are dealt
    to the FIRST ranks (one more to the first ``units % world_size``), so with
    fewer units than ranks the empty shards are the trailing ones -- they pass
    the halo and the carried state through."""
    align = samples_per_buffer * total_decimation // math.gcd(samples_per_buffer, total_decimation)
    units = -(-num_samples // align)
    per, extra = divmod(units, world_size)
    bounds = [0]
    for r in range(world_size - 1):
        bounds.append(min(num_samples, bounds[-1] + (per + (1 if r < extra else 0)) * align))
    bounds.append(num_samples)
    return bounds


@dataclass
class ShardResult:
    msg_samples: np.ndarray     # uint64, GLOBAL decimated index of OUTPUT_READY
    payloads: np.ndarray        # uint8 [n, payload_bytes]
    rounds: int                 # state-exchange rounds that changed something


def _to_tensor(buf: bytes, device):
    import torch
    t = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
    return t.to(device) if device is not None else t


def demodulate_sharded(engine, *, d_iq_ptr: int, num_local_samples: int, tail_samples: np.ndarray,
                       decimated_offset: int, group=None, comm_device=None,
                       max_rounds: int = 64) -> ShardResult:
    """Run the sharded protocol on this rank.

    engine            object with ``halo_samples``, ``shard_begin(ptr, n, halo,
                      last, state_in) -> (result, state_out)`` and
                      ``shard_refine(state_in) -> (result, state_out)``;
                      ``state_out`` objects expose ``bytes(state)`` and can be
                      rebuilt with ``engine.state_from_bytes``.
    tail_samples      the last ``halo_samples`` int16 I,Q samples of this shard
                      (host array, or a tensor on ``comm_device``: then the halo
                      never leaves the device) -- what the next rank needs as
                      FIR history.  Shorter when the shard is.
    decimated_offset  global decimated index of this shard's first output.
    comm_device       device for the tiny exchange tensors (``None`` = CPU for
                      gloo; ``torch.device('cuda', i)`` for nccl/RCCL).
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    H = int(engine.halo_samples)

    # ---- 1. halo: neighbour send/recv ---------------------------------------------
    # tail_samples / the received halo stay where they are (device tensors with nccl): nothing is
    # staged through the host.  A shard shorter than the halo (few alignment units, long filters)
    # cannot answer before it has heard from its predecessor: its tail is the end of
    # [received halo | own samples]; every other rank sends and receives at once.
    def as_tensor(x):
        if hasattr(x, "data_ptr"):
            return x.reshape(-1)
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.int16).reshape(-1).copy())
        return t.to(comm_device) if comm_device is not None else t

    def peer(r):
        return dist.get_global_rank(group, r) if group else r

    halo = None
    if H > 0 and world > 1:
        own = as_tensor(tail_samples)[-2 * min(H, num_local_samples):] if num_local_samples else None
        recv_t = torch.empty(2 * H, dtype=torch.int16, device=comm_device if comm_device is not None else "cpu") \
            if rank > 0 else None
        if num_local_samples >= H:
            reqs = []
            if rank + 1 < world:
                reqs.append(dist.isend(own.contiguous(), dst=peer(rank + 1), group=group))
            if recv_t is not None:
                reqs.append(dist.irecv(recv_t, src=peer(rank - 1), group=group))
            for q in reqs:
                q.wait()
        else:
            if recv_t is not None:
                dist.recv(recv_t, src=peer(rank - 1), group=group)
            if rank + 1 < world:
                before = recv_t if recv_t is not None else torch.zeros(2 * H, dtype=torch.int16,
                                                                         device=comm_device if comm_device is not None else "cpu")
                send_t = torch.cat([before, own.to(before.device)])[-2 * H:] if own is not None else before
                dist.send(send_t.contiguous(), dst=peer(rank + 1), group=group)
        halo = recv_t
        if halo is not None and getattr(halo, "is_cuda", False):
            # With nccl / RCCL `wait()` and `recv` only order torch's CURRENT stream behind the communication
            # stream; they neither block the host nor say anything to the receiver's own (non-blocking) HIP
            # stream, on which shard_begin copies the halo and runs the front end.  The halo must have landed
            # before that stream may read it.
            torch.cuda.current_stream(halo.device).synchronize()

    # ---- 2. speculative pass ----------------------------------------------------------
    result, out = engine.shard_begin(d_iq_ptr, num_local_samples, halo, rank == world - 1, None)

    # ---- 3. carried-state fix-point ------------------------------------------------------
    # ONE collective per round: every rank sees every shard's outgoing state, so "nothing changed
    # anywhere" is a comparison of this round's gathered states with the last round's (on the
    # communication device) -- no second collective for a flag.  Only the predecessor's 64 bytes come
    # to the host, and only when they differ from what this shard was run with.
    my_in: Optional[bytes] = None
    prev_all = None
    rounds = 0
    for _ in range(max_rounds + 1):
        mine = _to_tensor(bytes(out), comm_device)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        all_states = torch.stack(gathered)
        if prev_all is not None:
            if torch.equal(all_states, prev_all):
                break               # every shard ran with its predecessor's final outgoing state
            rounds += 1
        prev_all = all_states
        if rank > 0:
            want = bytes(gathered[rank - 1].cpu().numpy().tobytes())
            if want != my_in:
                # (the first round always lands here: the speculative pass
                #  assumed a state, now it learns the predecessor's)
                my_in = want
                result, out = engine.shard_refine(engine.state_from_bytes(want))
    else:
        raise RuntimeError("carried-state exchange did not converge")

    samples = np.asarray(result.msg_samples, dtype=np.uint64) + np.uint64(decimated_offset)
    return ShardResult(samples, np.asarray(result.payloads), rounds)


def gather_messages(local: ShardResult, group=None) -> Optional[ShardResult]:
    """Rank 0 receives every shard's messages, in capture order."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    box = [None] * world if rank == 0 else None
    dist.gather_object((local.msg_samples, local.payloads, local.rounds), box, dst=0, group=group)
    if rank != 0:
        return None
    samples = np.concatenate([b[0] for b in box]) if box else np.zeros(0, np.uint64)
    pays = np.concatenate([b[1] for b in box]) if box else np.zeros((0, 0), np.uint8)
    return ShardResult(samples, pays, max(b[2] for b in box))
