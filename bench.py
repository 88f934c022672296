#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated (SC16Q11 -> bits/messages) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload capture|batch|sharded]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload `capture` (default; BASELINE.json north_star: "a 16 GiB synthetic
SC16Q11 stream through the fs32_fs4 FIR + p3l-nexa2012 decoder"): synthetic
captures of 4 294 967 296 samples (recipe of SURVEY.md 8(d), one seed per
capture) resident in HBM, fs32_fs4.json FIR, p3l-nexa2012 decoder, threshold
0.1, 8192 samples per buffer, 3 Msps.  A step = one pass of the whole hot path
over one capture through the C ABI (ookd_rx_submit_device + ookd_rx_wait):
unpack + FIR + threshold + bit packing, edge extraction, symbol state machine,
decoded messages back in host memory.  By default three rx contexts are in
flight, each on its OWN capture: step k+1 is submitted before step k is waited
for, so its HBM-bound front end runs beside the latency-bound state machine of
the steps before.  With N > 1 every rank demodulates its own captures
(independent captures shard with no data-path collective: weak scaling); the
value is the whole-job aggregate.

Other workloads (BASELINE.json configs[3] / [4], per rank): `batch` = 128
independent captures of 2^24 samples in one batched call; `sharded` = one
capture cut into WORLD_SIZE shards, halo + carried state machine state
exchanged over torch.distributed (nccl = RCCL on the GPU box).

One JSON line is printed by rank 0, with `roofline` for the dominant kernel
(fused FIR front end, timed with HIP events on its own stream inside the
library), `single_context` (the same steps strictly one after the other),
`worst_case` (every window filtered: no quiet shortcut) and, at N = 1,
`cpu_baseline`: the CPU oracle (plain-C restatement of the reference path, 1
thread like the reference) timed on this host on a bounded slice of the same
capture -- also used as a parity check -- and the same code on all host cores.

`python bench.py --gpus N` without a launcher starts the N ranks itself (fresh
processes, before anything touches a GPU) and fails if the box has fewer
devices.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SAMPLES = 1 << 32                 # 16 GiB of SC16Q11: the north_star capture
BATCH_CAPTURES, BATCH_SAMPLES = 128, 1 << 24        # configs[3], one GPU's share
SEED_BASE = 0x00C0FFEE
RATE = 3000000
SPB = 8192
THRESHOLD = 0.1
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3            # vector fp32 peak
BYTES_PER_SAMPLE = 4.125            # 4 B read + 1/8 B written per decimated sample (D = 1)
FIR_FLOP_PER_SAMPLE = 128.0         # 32 real taps x (re, im) x (mul + add)
CPU_SLICE = 1 << 28                 # 1 GiB slice for the CPU baseline: ~5 s of one host core


def golden(kind, name):
    return os.path.join(ROOT, "tests", "golden", kind, name + ".json")


def spawn_ranks(args):
    """`--gpus N` without a launcher: N fresh worker processes, one per GPU.  Nothing in this
    process has touched a GPU (device_count() does not initialise one)."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not args.allow_shared_gpu:
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s); refusing to report fewer ranks than asked for "
                         "(--allow-shared-gpu: rehearse the multi-rank path with ranks sharing a GPU, gloo)"
                         % (args.gpus, ndev))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % max(1, ndev)), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("capture", "batch", "sharded"), default="capture")
    ap.add_argument("--samples", type=int, default=0, help="samples per capture / per shard (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true", help="skip the single_context / worst_case runs")
    ap.add_argument("--exact", action="store_true", help="unfused reference-order FIR everywhere")
    ap.add_argument("--filter", default="fs32_fs4", help=argparse.SUPPRESS)   # experiments only
    ap.add_argument("--contexts", type=int, default=3,
                    help="rx contexts (= HIP streams) with a capture in flight, each on its own capture: the memory-bound "
                         "front end of one step runs beside the state machine of the steps before; 1 = strictly one "
                         "step after the other")
    ap.add_argument("--no-quiet-skip", action="store_true",
                    help="filter every window, even those provably below the threshold (worst case)")
    ap.add_argument("--backend", default="", help="torch.distributed backend (default: nccl = RCCL; gloo when ranks share a GPU)")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="rehearsal: let --gpus N start N ranks on a box with fewer GPUs (the line says so; never a scaling number)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)               # does not return
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import ookiedokie_amd as ok

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    ndev = torch.cuda.device_count()
    shared_gpu = ndev < world               # rehearsal only: more ranks than GPUs on this box
    local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = args.backend or ("gloo" if shared_gpu else "nccl")     # RCCL refuses two ranks on one device
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    ok.lib()
    flt = ok.Filter.load(golden("filters", args.filter)) if args.filter != "none" else None
    decim = flt.total_decimation if flt else 1
    dev = ok.Device.load(golden("devices", "p3l-nexa2012"), RATE // decim)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes as C
    L = ok.lib()
    submit, wait, get_stats = L.ookd_rx_submit_device, L.ookd_rx_wait, L.ookd_rx_get_stats

    gate = ok.FrontGate()           # the contexts of this process take turns for their front ends

    def receiver(n, caps=1, **kw):
        return ok.Receiver(flt, dev, max_samples=n, max_captures=caps, threshold=THRESHOLD, samples_per_buffer=SPB,
                           hip_device=local_rank, exact_fir=args.exact, front_gate=gate, **kw)

    def synth_capture(n, seed):
        syn = ok.Synth(dev, n, seed=seed, sample_rate=RATE)
        buf = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
        syn.fill_device(buf.data_ptr(), hip_device=local_rank)
        return buf, syn

    def timed_steps(handles, ptrs, ncaps, n, stride, steps):
        """K steps over the contexts round-robin; returns seconds and the per-step kernel / device times."""
        fir_ms, dev_ms = [], []
        launches = [1]
        st = ok.RxStats()
        st_ref = C.byref(st)
        nctx = len(handles)

        def finish(h):
            if wait(h) != 0:
                raise SystemExit("ookd_rx_wait failed: " + ok.last_error())
            get_stats(h, st_ref)
            fir_ms.append(st.fir_kernel_ms)
            dev_ms.append(st.total_device_ms)
            launches[0] = max(1, st.front_launches)

        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            h = handles[k % nctx]
            if k >= nctx:
                finish(h)                   # step k - nctx ran on this context
            if submit(h, ptrs[k % nctx], ncaps, n, stride) != 0:
                raise SystemExit("ookd_rx_submit_device failed: " + ok.last_error())
        for k in range(max(0, steps - nctx), steps):
            finish(handles[k % nctx])
        barrier()
        return time.perf_counter() - t0, fir_ms, dev_ms, launches[0]

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    nctx = max(1, args.contexts)
    out = None

    def front_kernel_name(filt, exact=False, quiet=True):
        """the kernel the library launches for this filter (csrc/kernels.hip: launch_front)"""
        if filt is None:
            return "nofir_bits_kernel (unpack+threshold+bitpack)"
        if filt.total_decimation == 1 and filt.num_stages == 1 and not exact and not os.environ.get("OOKD_FIR_VALU"):
            return "fir1_mfma_kernel (unpack + split-fp16 Toeplitz FIR on the matrix cores + threshold + bitpack)"
        if filt.num_stages == 1:
            return "fir1_bits_kernel (unpack+FIR+threshold+bitpack)"
        if filt.num_stages == 2 and filt.total_decimation == 4 and not exact and not os.environ.get("OOKD_FIR_VALU"):
            return ("fir2_mfma_kernel (unpack + the two decimate-by-2 stages folded into one decimate-by-4 Toeplitz FIR on "
                    "the matrix cores + threshold + bitpack)")
        if filt.num_stages == 2:
            return "fir2_bits_kernel (unpack + two decimating FIR stages + threshold + bitpack)"
        return "fir_generic_kernel"

    def sub_record(what, filt, device, ptrs_, n_, ncaps_=1, stride_=None, contexts=(3, 1), steps=10, flop_per_sample=None,
                   **rkw):
        """One more workload through the same timed loop (its own contexts, its own short warm-up):
        ms per step with `contexts[0]` captures in flight and strictly one after the other."""
        stride_ = stride_ or n_
        rec = {"what": what}
        g2 = ok.FrontGate()
        mk = lambda: ok.Receiver(filt, device, max_samples=n_, max_captures=ncaps_, threshold=THRESHOLD,
                                 samples_per_buffer=SPB, hip_device=local_rank, front_gate=g2, **rkw)
        for nc in contexts:
            rr = [mk() for _ in range(nc)]
            hh = [r._h for r in rr]
            pp = [C.c_void_p(ptrs_[i % len(ptrs_)]) for i in range(nc)]
            last = None
            for i in range(max(3, nc)):
                last = rr[i % nc].rx_device(ptrs_[i % len(ptrs_)], n_, num_captures=ncaps_, stride=stride_)
            e, f, d, nl = timed_steps(hh, pp, ncaps_, n_, stride_, steps)
            kms = float(np.mean(f))
            dec_ = filt.total_decimation if filt else 1
            bps = 4.0 + 0.125 / dec_
            key = "contexts_%d" % nc
            rec[key] = {"ms_per_step": round(e / steps * 1e3, 4),
                        "value": round(float(n_) * ncaps_ * steps / e / 1e6, 1), "unit": "Msamples/s",
                        "frac_of_hbm_read_roofline": round(float(n_) * ncaps_ * steps / e / 1e6 / (HBM_PEAK_GBS * 1e3 / 4.0), 4),
                        "kernel_ms": round(kms, 4),
                        "kernel_frac_of_hbm_peak": round(bps * n_ * ncaps_ / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "front_launches": nl}
            if flop_per_sample:
                rec[key]["fir_tflops"] = round(flop_per_sample * n_ * ncaps_ / (kms * 1e-3) / 1e12, 2)
            rec["messages_per_step"] = int(len(last.msg_samples))
            rec["edges_per_step"] = int(last.stats["num_edges"])
            rec["fsm_path"] = int(last.stats["fsm_path"])
            rec["guard_recomputes"] = int(last.stats["guard_recomputes"])
            for r in rr:
                r.close()
        g2.close()
        rec["kernel"] = front_kernel_name(filt)
        return rec

    # =====================================================================================================
    if args.workload == "sharded":
        # configs[4]: one capture cut into WORLD_SIZE contiguous shards, halo + carried state exchanged
        from ookiedokie_amd.distributed import demodulate_sharded
        n = args.samples or (1 << 33)               # per shard: configs[4] = 256 GiB over 8 GPUs = 2^33 samples (32 GiB) each
        total = n * world
        syn = ok.Synth(dev, total, seed=SEED_BASE + 5, sample_rate=RATE)
        shard = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
        syn.fill_device(shard.data_ptr(), first=rank * n, count=n, hip_device=local_rank)
        torch.cuda.synchronize()
        rx = receiver(n)
        H = int(rx.halo_samples)
        tail = shard[2 * (n - H):2 * n] if H else shard[:0]
        comm = torch.device("cuda", local_rank) if (dist is not None and dist.get_backend() == "nccl") else None
        if comm is None:
            tail = tail.cpu().numpy()

        def one_step():
            if dist is None:
                return rx.shard_begin(shard.data_ptr(), n, None, True, None)[0]
            return demodulate_sharded(rx, d_iq_ptr=shard.data_ptr(), num_local_samples=n, tail_samples=tail,
                                      decimated_offset=rank * n // decim, comm_device=comm)
        res = None
        for _ in range(max(1, args.warmup)):
            res = one_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = one_step()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        # ---- checker (untimed): the shards' messages, gathered in capture order, against the whole capture --------
        allmsgs = res
        if dist is not None:
            from ookiedokie_amd.distributed import gather_messages
            allmsgs = gather_messages(res)
        check = {}
        if rank == 0:
            ms = np.asarray(allmsgs.msg_samples, dtype=np.uint64)
            pays = np.asarray(allmsgs.payloads)
            check["messages"] = int(ms.size)
            check["messages_in_increasing_order"] = bool((np.diff(ms.astype(np.int64)) > 0).all()) if ms.size > 1 else True
            check["last_message_sample"] = int(ms[-1]) if ms.size else 0
            free, _ = torch.cuda.mem_get_info()
            if world > 1 and 4 * total + (2 << 30) < free and total <= (1 << 32):
                # the whole capture fits beside the shard: decode it in one piece and compare
                whole = torch.empty(2 * total + 64, dtype=torch.int16, device="cuda")
                syn.fill_device(whole.data_ptr(), hip_device=local_rank)
                wrx = ok.Receiver(flt, dev, max_samples=total, threshold=THRESHOLD, samples_per_buffer=SPB, hip_device=local_rank)
                wres = wrx.rx_device(whole.data_ptr(), total)
                check["sharded_equals_whole"] = bool(list(wres.msg_samples) == [int(x) for x in ms]
                                                     and (np.asarray(wres.payloads) == pays).all())
                wrx.close()
                del whole
            else:
                # too large for one GPU: every decoded payload was transmitted, in transmission order, and nearly all were
                sent = [syn.message(i)[1] for i in range(syn.num_messages)]
                j, okk = 0, True
                for pl in pays:
                    while j < len(sent) and sent[j] != bytes(pl):
                        j += 1
                    if j >= len(sent):
                        okk = False
                        break
                    j += 1
                check["sharded_equals_whole"] = None
                check["decoded_are_sent_in_order"] = bool(okk and len(pays) >= 0.8 * (len(sent) - 2))
        if rank == 0:
            st = rx.stats()
            out = {
                "metric": "IQ Msamples/s demodulated (SC16Q11->bits)",
                "value": round(float(total) * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "configs[4]-shaped: ONE capture of %d samples cut into %d contiguous shards of %d "
                                       "samples, overlap-save halo (%d samples) + carried state machine state exchanged "
                                       "between neighbours, fs32_fs4 FIR, p3l-nexa2012 state machine" % (total, world, n, H),
                           "samples_per_shard": n, "shards": world, "halo_samples": H, "filter": args.filter,
                           "device": "p3l-nexa2012", "backend": backend or "none",
                           "state_exchange_rounds": int(getattr(res, "rounds", 0)),
                           "rehearsal_ranks_share_a_gpu": bool(shared_gpu),
                           "parallelism": "contiguous shards, neighbour halo send/recv + all-gather of a 64-byte state",
                           **check},
                "roofline": {"kernel": front_kernel_name(flt, args.exact), "bound": "hbm",
                             "achieved": round(BYTES_PER_SAMPLE * n / (st["fir_kernel_ms"] * 1e-3) / 1e9, 1),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(BYTES_PER_SAMPLE * n / (st["fir_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "traffic": None, "avg_kernel_ms": round(st["fir_kernel_ms"], 4)},
            }
    # =====================================================================================================
    else:
        if args.workload == "batch":
            n, ncaps = args.samples or BATCH_SAMPLES, BATCH_CAPTURES
            stride = n + 64
        else:
            n, ncaps = args.samples or N_SAMPLES, 1
            stride = n
        # ---- synthetic captures generated directly in HBM: one (set) per context ---------------------------------
        bufs, syns = [], []
        for c in range(nctx):
            if ncaps == 1:
                b, s = synth_capture(n, SEED_BASE + 2 + 16 * rank + c)
            else:
                b = torch.empty(2 * stride * ncaps + 64, dtype=torch.int16, device="cuda")
                for i in range(ncaps):
                    s = ok.Synth(dev, n, seed=SEED_BASE + 1000 + (rank * nctx + c) * ncaps + i, sample_rate=RATE)
                    s.fill_device(b.data_ptr() + 4 * stride * i, hip_device=local_rank)
            bufs.append(b)
            syns.append(s)
        torch.cuda.synchronize()
        kw = dict(message_capacity=1 << 18) if ncaps > 1 else {}
        rxs = [receiver(n, ncaps, quiet_skip=not args.no_quiet_skip, **kw) for _ in range(nctx)]
        handles = [r._h for r in rxs]
        ptrs = [C.c_void_p(b.data_ptr()) for b in bufs]
        for i in range(max(args.warmup, nctx)):
            rxs[i % nctx].rx_device(bufs[i % nctx].data_ptr(), n, num_captures=ncaps, stride=stride)
        elapsed, fir_ms, dev_ms, nlaunch = timed_steps(handles, ptrs, ncaps, n, stride, args.steps)
        elapsed = max_over_ranks(elapsed)
        res = rxs[(args.steps - 1) % nctx].result()

        quiet_frac = 0.0
        single = worst = None
        if rank == 0 and not args.no_quiet_skip:
            # untimed diagnostic pass: how many 512-output windows took the quiet shortcut
            cnt = receiver(n, ncaps, count_quiet=True, **kw)
            st = cnt.rx_device(bufs[0].data_ptr(), n, num_captures=ncaps, stride=stride).stats
            quiet_frac = st["quiet_waves"] / max(1, st["total_waves"])
            cnt.close()
        if rank == 0 and world == 1 and not args.no_sub_records:
            # ---- the same steps strictly one after the other (one context) ---------------------------------
            k1 = max(3, min(args.steps, 20))
            timed_steps(handles[:1], ptrs[:1], ncaps, n, stride, 2)        # (untimed: the mode's own warm-up)
            e1, f1, d1, _ = timed_steps(handles[:1], ptrs[:1], ncaps, n, stride, k1)
            single = {"ms_per_step": round(e1 / k1 * 1e3, 4), "value": round(float(n) * ncaps * k1 / e1 / 1e6, 1),
                      "unit": "Msamples/s", "steps": k1, "kernel_ms": round(float(np.mean(f1)), 4),
                      "kernel_frac_of_hbm_peak": round(BYTES_PER_SAMPLE * n * ncaps / (float(np.mean(f1)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "device_ms_per_step": round(float(np.mean(d1)), 4),
                      "frac_of_hbm_read_roofline": round(float(n) * ncaps * k1 / e1 / 1e6 / (HBM_PEAK_GBS * 1e3 / 4.0), 4)}
            # ---- worst case: every window filtered (a carrier that never drops) ---------------------------------
            if not args.no_quiet_skip and flt is not None:
                wrx = receiver(n, ncaps, quiet_skip=False, **kw)
                wrx.rx_device(bufs[0].data_ptr(), n, num_captures=ncaps, stride=stride)
                k2 = 3
                e2, f2, _, _ = timed_steps([wrx._h], ptrs[:1], ncaps, n, stride, k2)
                kms = float(np.mean(f2))
                tf = FIR_FLOP_PER_SAMPLE * n * ncaps / (kms * 1e-3) / 1e12
                worst = {"what": "OOKD_RX_NO_QUIET_SKIP: the filter runs on every window", "ms_per_step": round(e2 / k2 * 1e3, 4),
                         "value": round(float(n) * ncaps * k2 / e2 / 1e6, 1), "unit": "Msamples/s", "kernel_ms": round(kms, 4),
                         "kernel_frac_of_hbm_peak": round(BYTES_PER_SAMPLE * n * ncaps / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "fir_tflops": round(tf, 2), "frac_of_fp32_valu_peak": round(tf / FP32_PEAK_TFLOPS, 4)}
                wrx.close()
        fallback = None
        if rank == 0 and world == 1 and not args.no_sub_records and ncaps == 1:
            # ---- what a capture costs when the scan form refuses it: the round form, on a 1 GiB slice ---------
            m = min(n, 1 << 28)
            frx = receiver(m, fsm_rounds=True)
            srx = receiver(m)
            for r_ in (frx, srx):
                r_.rx_device(bufs[0].data_ptr(), m)
            t0 = time.perf_counter()
            frx.process_device(bufs[0].data_ptr(), m)
            t_rounds = time.perf_counter() - t0
            t0 = time.perf_counter()
            srx.process_device(bufs[0].data_ptr(), m)
            t_scan = time.perf_counter() - t0
            fallback = {"what": "the state machine's round form (what a capture costs when the scan form refuses it), "
                                "one call over the first %d samples" % m,
                        "rounds_ms": round(t_rounds * 1e3, 4), "scan_ms": round(t_scan * 1e3, 4),
                        "fix_point_rounds": int(frx.stats()["fsm_iterations"])}
            frx.close()
            srx.close()
        configs = None
        if (rank == 0 and world == 1 and not args.no_sub_records and args.workload == "capture" and n == N_SAMPLES
                and args.filter == "fs32_fs4" and not args.no_quiet_skip and not args.exact):
            # ---- the other single-GPU configurations of BASELINE.json, each through the same timed loop --------
            configs = {}
            p0 = [b.data_ptr() for b in bufs]
            configs["config1"] = sub_record(
                "configs[1]: 1 GiB synthetic SC16Q11 capture (268435456 samples), fs32_fs4 FIR, p3l-nexa2012, 1 MI355X",
                flt, dev, p0, 1 << 28, flop_per_sample=FIR_FLOP_PER_SAMPLE)
            configs["dec4"] = sub_record(
                "backend default filter fs128_fs16_dec4 (16 taps / 2, 32 taps / 2: SURVEY.md 8(f) row f1), 16 GiB capture, p3l-nexa2012",
                ok.Filter.load(golden("filters", "fs128_fs16_dec4")), ok.Device.load(golden("devices", "p3l-nexa2012"), RATE // 4),
                p0, n, contexts=(3,), steps=8)
            # configs[2]: 255 real taps = float32(hamming-windowed sinc, cutoff Fs/64, unity DC gain), unknown-remote1
            import tempfile
            kk = np.arange(255) - 127
            hh = np.sinc(kk / 32.0) * np.hamming(255)
            hh = hh / hh.sum()
            tdir = tempfile.mkdtemp()
            with open(os.path.join(tdir, "sinc255.json"), "w") as f:
                json.dump({"filter": {"stages": [{"decimation": 1, "taps": list(hh)}]}}, f)
            f255 = ok.Filter.load(os.path.join(tdir, "sinc255.json"))
            dev2 = ok.Device.load(golden("devices", "unknown-remote1"), RATE)
            syn2 = ok.Synth(dev2, n, seed=SEED_BASE + 3, sample_rate=RATE)
            cap2 = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
            syn2.fill_device(cap2.data_ptr(), hip_device=local_rank)
            torch.cuda.synchronize()
            configs["config2"] = sub_record(
                "configs[2]: 16 GiB synthetic SC16Q11 capture (4294967296 samples), 255-tap synthetic FIR (hamming-windowed "
                "sinc, cutoff Fs/64), unknown-remote1 decoder, 1 MI355X; the contexts share one resident capture",
                f255, dev2, [cap2.data_ptr()], n, contexts=(3, 1), steps=6, flop_per_sample=1020.0)
            del cap2
            # configs[3], one GPU's share: 128 captures of 2^24 samples per batched call
            bstride = BATCH_SAMPLES + 64
            bbuf = torch.empty(2 * bstride * BATCH_CAPTURES + 64, dtype=torch.int16, device="cuda")
            for i in range(BATCH_CAPTURES):
                sb = ok.Synth(dev, BATCH_SAMPLES, seed=SEED_BASE + 1000 + i, sample_rate=RATE)
                sb.fill_device(bbuf.data_ptr() + 4 * bstride * i, hip_device=local_rank)
            torch.cuda.synchronize()
            configs["batch"] = sub_record(
                "configs[3], one GPU's share: %d independent captures of %d samples (64 MiB) in ONE batched call per step, "
                "fs32_fs4 FIR, p3l-nexa2012; the contexts share one resident batch" % (BATCH_CAPTURES, BATCH_SAMPLES),
                flt, dev, [bbuf.data_ptr()], BATCH_SAMPLES, ncaps_=BATCH_CAPTURES, stride_=bstride, contexts=(3,), steps=8,
                flop_per_sample=FIR_FLOP_PER_SAMPLE, message_capacity=1 << 18)
            del bbuf
        if rank == 0:
            total_samples = float(n) * ncaps * args.steps * world
            value = total_samples / elapsed / 1e6
            fir_avg_ms = float(np.mean(fir_ms))
            achieved_gbs = BYTES_PER_SAMPLE * n * ncaps / (fir_avg_ms * 1e-3) / 1e9
            # flops actually executed: quiet windows skip the filter altogether
            fir_tflops = FIR_FLOP_PER_SAMPLE * n * ncaps * (1.0 - quiet_frac) / (fir_avg_ms * 1e-3) / 1e12
            traffic = None
            prof = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(prof) and args.workload == "capture":
                try:
                    with open(prof) as f:
                        tj = json.load(f)
                    if tj.get("samples_per_launch"):
                        traffic = tj["hbm_bytes_per_launch"] * (float(n) * ncaps / nlaunch / tj["samples_per_launch"])
                except Exception:
                    traffic = None
            if args.workload == "batch":
                wl = ("configs[3], one GPU's share: %d independent synthetic SC16Q11 captures of %d samples (64 MiB) in ONE "
                      "batched call per step, fs32_fs4 FIR, p3l-nexa2012 state machine" % (ncaps, n))
            else:
                wl = ("north_star: %d GiB synthetic SC16Q11 capture per step (%d samples), fs32_fs4 FIR, p3l-nexa2012 "
                      "state machine, 1 MI355X per rank" % (n * 4 >> 30, n))
            out = {
                "metric": "IQ Msamples/s demodulated (SC16Q11->bits)",
                "value": round(value, 1),
                "unit": "Msamples/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f32",
                "data": "synthetic",
                "frac_of_hbm_read_roofline": round(value / world / (HBM_PEAK_GBS * 1e3 / 4.0), 4),
                "config": {
                    "workload": wl,
                    "samples_per_capture": n, "captures_per_step": ncaps, "filter": args.filter,
                    "device": "p3l-nexa2012", "sample_rate": RATE, "threshold": THRESHOLD,
                    "samples_per_buffer": SPB, "fir_mode": "exact" if args.exact else "fp16-split MFMA, exact products + guard band (bits exact)",
                    "parallelism": "independent captures per rank, no collective",
                    "backend": backend or "none",
                    "messages_per_step": int(len(res.msg_samples)),
                    "edges_per_step": int(res.stats["num_edges"]),
                    "fsm_path": int(res.stats["fsm_path"]),
                    "guard_recomputes": int(res.stats["guard_recomputes"]),
                    "contexts_in_flight": nctx,
                    "captures_resident": nctx,
                    "quiet_shortcut": not args.no_quiet_skip,
                    "quiet_window_fraction": round(quiet_frac, 4),
                },
                "roofline": {
                    # the front end of a step goes out as `launches_per_step` grid launches of the same
                    # kernel; the library times first start -> last end (HIP events riding on the
                    # dispatches), so one launch lasts that / launches and moves bytes / launches
                    "kernel": front_kernel_name(flt, args.exact),
                    "bound": "hbm",
                    "achieved": round(achieved_gbs, 1),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "traffic_note": "PMC bytes (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) of a --contexts 1 run, per launch: "
                                    "profiles/traffic.json; the time above is with %d contexts in flight" % nctx,
                    "launches_per_step": nlaunch,
                    "avg_kernel_ms": round(fir_avg_ms / nlaunch, 4),
                    "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * n * ncaps / nlaunch,
                    "front_end_ms_per_step": round(fir_avg_ms, 4),
                    # the 32-tap FIR sits above the fp32 ridge (SURVEY.md hard part 1); flops
                    # of the windows that really ran the filter:
                    "fir_tflops": round(fir_tflops, 2),
                    "frac_of_fp32_valu_peak": round(fir_tflops / FP32_PEAK_TFLOPS, 4),
                },
                # first kernel start -> last kernel end of one capture (its latency on the device;
                # with several contexts in flight consecutive captures overlap, so ms_per_step is smaller)
                "device_ms_per_step": round(float(np.mean(dev_ms)), 4),
                "single_context": single,
                "worst_case": worst,
                "fallback_path": fallback,
                "configs": configs,
            }

        # ---- CPU baseline: the oracle on this host's cores (rank 0, N = 1 only) -------------
        if rank == 0 and world == 1 and args.workload == "capture" and not args.no_cpu_baseline:
            import oracle as O
            O.build()
            m = min(CPU_SLICE, n)
            iq = bufs[0][:2 * m].cpu().numpy()
            ofir = O.load_filter_json(golden("filters", args.filter)) if flt else None
            odev, _ = O.load_device_json(golden("devices", "p3l-nexa2012"), RATE // decim)
            t1 = time.perf_counter()
            want = O.rx(iq, ofir, THRESHOLD, odev, SPB)
            cpu_s = time.perf_counter() - t1
            # checker: the GPU result over the same slice must be identical
            chk = receiver(m)
            got = chk.rx_device(bufs[0].data_ptr(), m)
            parity = (list(got.msg_samples) == list(want.msg_samples)
                      and bool((got.payloads == want.payloads).all())
                      and got.stats["num_errors"] == len(want.err_samples))
            chk.close()
            # the same code on every core this process may use, one independent slice per thread
            # (the reference is single-threaded, users run one capture per core)
            from concurrent.futures import ThreadPoolExecutor
            try:
                ncores = len(os.sched_getaffinity(0))
            except AttributeError:
                ncores = os.cpu_count() or 1
            nthreads = max(1, min(ncores, 512))
            per = max(SPB, (min(1 << 24, 4 * m // nthreads)) // SPB * SPB)          # <= 2^24 samples per thread
            all_cores = None
            if nthreads > 1:
                slices = [iq[2 * ((i * per) % (m - per + 1) // SPB * SPB):][:2 * per] for i in range(nthreads)]
                t2 = time.perf_counter()
                with ThreadPoolExecutor(nthreads) as pool:          # the ctypes call releases the GIL
                    list(pool.map(lambda x: O.rx(x, ofir, THRESHOLD, odev, SPB), slices))
                all_s = time.perf_counter() - t2
                all_cores = {"value": round(nthreads * per / all_s / 1e6, 2), "unit": "Msamples/s", "cores": nthreads,
                             "sample": "%d slices of %d samples of the same capture, one thread per core this process "
                                       "may run on, %.1f s" % (nthreads, per, all_s)}
            cpu_model = "unknown"
            try:
                with open("/proc/cpuinfo") as f:
                    for line in f:
                        if line.startswith("model name"):
                            cpu_model = line.split(":", 1)[1].strip()
                            break
            except OSError:
                pass
            out["cpu_baseline"] = {
                "value": round(m / cpu_s / 1e6, 2),
                "unit": "Msamples/s",
                "cores": 1,
                "kind": "port",
                "sample": "first %d samples (%d MiB) of the first capture, oracle/ook_oracle.c "
                          "(scalar C restatement, -O3 no FMA, 1 thread like the reference), %.1f s"
                          % (m, m * 4 >> 20, cpu_s),
                "host_cpu": cpu_model,
                "host_cores": os.cpu_count(),
                "gpu_matches_oracle_on_sample": parity,
                "all_cores": all_cores,
            }
            if not parity:
                print(json.dumps(out))
                raise SystemExit("PARITY FAILURE: GPU result differs from the oracle on the CPU sample")

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
