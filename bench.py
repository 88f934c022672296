#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated (SC16Q11 -> bits/messages) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted
on): one 1 GiB synthetic SC16Q11 capture (268 435 456 samples, recipe of
SURVEY.md 8(d), seed base+2+rank) resident in HBM, fs32_fs4.json FIR,
p3l-nexa2012 decoder, threshold 0.1, 8192 samples per buffer, 3 Msps.
A step = one pass of the whole hot path over that capture: unpack + FIR +
threshold + bit packing, edge extraction, symbol state machine, decoded
messages back in host memory.  With N > 1 every rank demodulates its own
capture (independent captures shard with no data-path collective: weak
scaling); the value is the whole-job aggregate.

One JSON line is printed by rank 0, with `roofline` for the dominant kernel
(fused FIR front end, timed with HIP events on its own stream inside the
library) and, at N = 1, `cpu_baseline`: the CPU oracle (plain-C restatement
of the reference path, 1 thread like the reference) timed on this host on a
bounded slice of the same capture -- also used as a final parity check.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SAMPLES = 1 << 28                 # 1 GiB of SC16Q11
SEED_BASE = 0x00C0FFEE
RATE = 3000000
SPB = 8192
THRESHOLD = 0.1
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3            # vector fp32 peak
BYTES_PER_SAMPLE = 4.125            # 4 B read + 1/8 B written per decimated sample (D = 1)
FIR_FLOP_PER_SAMPLE = 128.0         # 32 real taps x (re, im) x (mul + add)
CPU_SLICE = 1 << 28             # the whole bench capture: ~5 s of one host core, and a full-size parity check


def golden(kind, name):
    return os.path.join(ROOT, "tests", "golden", kind, name + ".json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples", type=int, default=N_SAMPLES, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact", action="store_true", help="unfused reference-order FIR everywhere")
    ap.add_argument("--filter", default="fs32_fs4", help=argparse.SUPPRESS)   # experiments only
    ap.add_argument("--contexts", type=int, default=3,
                    help="rx contexts (= HIP streams) with a capture in flight: with 2 or 3 the memory-bound front "
                         "end of one step runs while the state machine of the steps before finishes (3 measured "
                         "best: +8 %% over 2, 4 is worse again); 1 = strictly one step after the other")
    ap.add_argument("--no-quiet-skip", action="store_true",
                    help="filter every window, even those provably below the threshold (worst case)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import ookiedokie_amd as ok

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    ndev = torch.cuda.device_count()
    shared_gpu = local_rank >= ndev         # rehearsal only: more ranks than GPUs on this box
    local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared_gpu or ndev < world:
            dist.init_process_group("gloo")         # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ok.lib()
    n = args.samples
    flt = ok.Filter.load(golden("filters", args.filter)) if args.filter != "none" else None
    dev = ok.Device.load(golden("devices", "p3l-nexa2012"), RATE // (flt.total_decimation if flt else 1))

    # ---- synthetic capture generated directly in HBM ----------------------------------
    syn = ok.Synth(dev, n, seed=SEED_BASE + 2 + rank, sample_rate=RATE)
    capture = torch.empty(2 * n + 64, dtype=torch.int16, device="cuda")
    syn.fill_device(capture.data_ptr(), hip_device=local_rank)
    torch.cuda.synchronize()

    nctx = max(1, args.contexts)
    rxs = [ok.Receiver(flt, dev, max_samples=n, threshold=THRESHOLD, samples_per_buffer=SPB,
                       hip_device=local_rank, exact_fir=args.exact, quiet_skip=not args.no_quiet_skip)
           for _ in range(nctx)]
    rx = rxs[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for i in range(max(args.warmup, nctx)):
        res = rxs[i % nctx].rx_device(capture.data_ptr(), n)
    fir_ms, dev_ms = [], []
    # One step = one capture through the C ABI: submit queues the whole hot path on the
    # context's stream, wait returns with the decoded messages in host memory.  With several
    # contexts step k+1 is submitted before step k is waited for, so its front end (HBM
    # bound) runs beside the state machine of step k (latency bound); every step is still
    # one complete pass over the capture, and all K complete inside the timed bracket.
    # The ctypes objects are bound once so the loop measures the library, not Python.
    import ctypes as C
    L = ok.lib()
    submit, wait, get_stats = L.ookd_rx_submit_device, L.ookd_rx_wait, L.ookd_rx_get_stats
    handles, ptr = [r._h for r in rxs], C.c_void_p(capture.data_ptr())
    st = ok.RxStats()
    st_ref = C.byref(st)

    def finish(h):
        if wait(h) != 0:
            raise SystemExit("ookd_rx_wait failed: " + ok.last_error())
        get_stats(h, st_ref)
        fir_ms.append(st.fir_kernel_ms)
        dev_ms.append(st.total_device_ms)

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        h = handles[k % nctx]
        if k >= nctx:
            finish(h)                   # step k - nctx ran on this context
        if submit(h, ptr, 1, n, n) != 0:
            raise SystemExit("ookd_rx_submit_device failed: " + ok.last_error())
    for k in range(max(0, args.steps - nctx), args.steps):
        finish(handles[k % nctx])
    barrier()
    elapsed = time.perf_counter() - t0
    res = rxs[(args.steps - 1) % nctx].result()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    quiet_frac = 0.0
    if rank == 0 and not args.no_quiet_skip:
        # untimed diagnostic pass: how many 1024-output windows took the quiet shortcut
        cnt = ok.Receiver(flt, dev, max_samples=n, threshold=THRESHOLD, samples_per_buffer=SPB,
                          hip_device=local_rank, exact_fir=args.exact, count_quiet=True)
        st = cnt.rx_device(capture.data_ptr(), n).stats
        quiet_frac = st["quiet_waves"] / max(1, st["total_waves"])
        cnt.close()
    if rank == 0:
        total_samples = float(n) * args.steps * world
        value = total_samples / elapsed / 1e6
        fir_avg_ms = float(np.mean(fir_ms))
        achieved_gbs = BYTES_PER_SAMPLE * n / (fir_avg_ms * 1e-3) / 1e9
        # flops actually executed: quiet windows skip the filter altogether
        fir_tflops = FIR_FLOP_PER_SAMPLE * n * (1.0 - quiet_frac) / (fir_avg_ms * 1e-3) / 1e12
        traffic = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                with open(prof) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
            "config": {
                "workload": "configs[1]: 1 GiB synthetic SC16Q11 capture per GPU, fs32_fs4 FIR, "
                            "p3l-nexa2012 state machine",
                "samples_per_capture": n, "captures_per_gpu": 1, "filter": args.filter,
                "device": "p3l-nexa2012", "sample_rate": RATE, "threshold": THRESHOLD,
                "samples_per_buffer": SPB, "fir_mode": "exact" if args.exact else "fma+guard-band",
                "parallelism": "independent captures per rank, no collective",
                "messages_per_capture": int(len(res.msg_samples)),
                "edges_per_capture": int(res.stats["num_edges"]),
                "fsm_rounds": int(res.stats["fsm_iterations"]),
                "guard_recomputes": int(res.stats["guard_recomputes"]),
                "contexts_in_flight": nctx,
                "quiet_shortcut": not args.no_quiet_skip,
                "quiet_window_fraction": round(quiet_frac, 4),
            },
            "roofline": {
                "kernel": "fir1_bits_kernel (unpack+FIR+threshold+bitpack)",
                "bound": "hbm",
                "achieved": round(achieved_gbs, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "avg_kernel_ms": round(fir_avg_ms, 4),
                "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * n,
                # the 32-tap FIR sits above the fp32 ridge (SURVEY.md hard part 1); flops
                # of the windows that really ran the filter:
                "fir_tflops": round(fir_tflops, 2),
                "frac_of_fp32_valu_peak": round(fir_tflops / FP32_PEAK_TFLOPS, 4),
            },
            # first kernel start -> last kernel end of one capture (its latency on the device;
            # with several contexts in flight consecutive captures overlap, so ms_per_step is smaller)
            "device_ms_per_step": round(float(np.mean(dev_ms)), 4),
        }

    # ---- CPU baseline: the oracle on this host's cores (rank 0, N = 1 only) -------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle as O
        O.build()
        m = min(CPU_SLICE, n)
        iq = capture[:2 * m].cpu().numpy()
        ofir = O.load_filter_json(golden("filters", args.filter)) if flt else None
        odev, _ = O.load_device_json(golden("devices", "p3l-nexa2012"), RATE // (flt.total_decimation if flt else 1))
        t1 = time.perf_counter()
        want = O.rx(iq, ofir, THRESHOLD, odev, SPB)
        cpu_s = time.perf_counter() - t1
        # checker: the GPU result over the same slice must be identical
        if m == n:
            got = res                   # the timed run's own result
        else:
            chk = ok.Receiver(flt, dev, max_samples=m, threshold=THRESHOLD, samples_per_buffer=SPB,
                              hip_device=local_rank, exact_fir=args.exact)
            got = chk.rx_device(capture.data_ptr(), m)
        parity = (list(got.msg_samples) == list(want.msg_samples)
                  and bool((got.payloads == want.payloads).all())
                  and got.stats["num_errors"] == len(want.err_samples))
        # the same code on every core this process may use, one independent slice per thread
        # (the reference is single-threaded, users run one capture per core)
        from concurrent.futures import ThreadPoolExecutor
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        nthreads = max(1, min(ncores, 16))              # a one-GPU box's share of the host
        per = m // nthreads // SPB * SPB
        all_cores = None
        if per > 0 and nthreads > 1:
            slices = [iq[2 * i * per:2 * (i + 1) * per] for i in range(nthreads)]
            t2 = time.perf_counter()
            with ThreadPoolExecutor(nthreads) as pool:          # the ctypes call releases the GIL
                list(pool.map(lambda x: O.rx(x, ofir, THRESHOLD, odev, SPB), slices))
            all_s = time.perf_counter() - t2
            all_cores = {"value": round(nthreads * per / all_s / 1e6, 2), "unit": "Msamples/s", "cores": nthreads,
                         "sample": "%d slices of %d samples of the same capture, one thread each, %.1f s"
                                   % (nthreads, per, all_s)}
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
            "sample": "first %d samples (%d MiB) of the same capture, oracle/ook_oracle.c "
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
