#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DWT backend (driver contract: one JSON line).

A "step" is one full forward transform (the non-allocating dwt!(y, x, wt, L) entry point) of one
synthetic array already resident in HBM:
    default workload  C3 = 2-D dwt, WT.db4 filter bank, 8192 x 8192 Float32, L = 13 (API default)
                      (BASELINE.json configs[2], the configuration the metric is quoted on)
With --gpus N > 1 the script runs one rank per GPU over RCCL: under torch.distributed.run (the driver's form) it
reads RANK / LOCAL_RANK / WORLD_SIZE; started plainly (`python bench.py --gpus N`) it re-launches ITSELF through
torch.distributed.run with N ranks.  Every rank transforms its own independent 8192 x 8192 array (a batch of N images
sharded one per GPU: weak scaling, no data-path collective; the filter taps are broadcast from rank 0 over
RCCL/xGMI before the timed region, as north_star prescribes).  value = whole-job Msamples/s = N * samples /
max-over-ranks time.  Beside it, `c5_batched` reports BASELINE.json configs[4]: the 65536-signal x 2^16 batch sharded
65536/N columns per rank (sharding.shard_range), aggregate Msamples/s and GB/s, checksum all-reduced.

Extra objects on the JSON line:
  roofline      dominant kernel (the level-1 launch of k_fwd2d_stream, which moves 8 B/sample):
                algorithmic bytes / average launch duration measured with HIP events on the launch
                stream around single-launch (L = 1) calls; peak = 8000 GB/s (MI355X HBM3E spec)
  cpu_baseline  the oracle (literal C restatement of the reference's loops, 1 thread -- the
                reference has no threading) timed on this host on a bounded sample

Other workloads (parity-test configs, not the headline): --workload c1|c2|c4|c5.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--levels", type=int, default=None, help="override L (default: maxtransformlevels)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--path", type=int, default=0, help="0 fast kernels, 1 generic kernels only")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other BASELINE configs")
    ap.add_argument("--no-c5", action="store_true", help="skip the sharded C5 batch object")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the 4-stream leg (profiling runs: overlapping launches skew per-kernel statistics)")
    ap.add_argument("--c5-signals", type=int, default=65536, help="total signals of the sharded C5 batch (BASELINE: 65536)")
    ap.add_argument("--stub-backend", default=None, help=argparse.SUPPRESS)   # tests only: 'gloo' = CPU ranks, stub transform
    return ap.parse_args()


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def respawn_command(args, argv):
    """`python bench.py --gpus N` started without a launcher: the torch.distributed.run command line that runs the same
    script with N ranks on this node (None when no re-launch is needed)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def make_workload(W, name, device, seed):
    """returns (label, x (device tensor), wt, default L, sample count, call(x) -> y, dtype tag)"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    WT = W.WT
    if name == "c3":
        x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).to(device).t()
        return "2-D dwt db4 filter 8192x8192 f32", x, W.wavelet(WT.db4), W.maxtransformlevels(x), "f32"
    if name == "c2":
        x = torch.randn(1 << 24, generator=g, dtype=torch.float32).to(device)
        return "1-D dwt db4 filter 2^24 f32", x, W.wavelet(WT.db4), 24, "f32"
    if name == "c4":
        x = torch.randn(1 << 24, generator=g, dtype=torch.float32).to(device)
        return "1-D dwt cdf9/7 lifting 2^24 f32", x, W.wavelet(WT.cdf97, WT.Lifting), 24, "f32"
    if name == "c1":
        x = torch.rand(1 << 20, generator=g, dtype=torch.float64).to(device)
        return "1-D dwt db2 filter 2^20 f64", x, W.wavelet(WT.db2), 20, "f64"
    if name == "c5":
        # per-GPU shard of the 65536 x 2^16 batch on 8 GPUs: 8192 signals of length 2^16
        x = torch.randn(8192, 1 << 16, generator=g, dtype=torch.float32).to(device).t()
        return "batched column-wise dwt db4, 8192 signals x 2^16 f32 per GPU", x, W.wavelet(WT.db4), 16, "f32"
    raise ValueError(name)


def stub_main(args, rank, world):
    """Launcher / reduction plumbing on CPU ranks (tests/test_bench_launch.py): gloo backend, the transform replaced by a
    copy.  Never a measurement: the line says so."""
    import torch.distributed as dist
    from wavelets_jl_amd import sharding
    dist.init_process_group(backend=args.stub_backend)
    cpu = torch.device("cpu")
    x = torch.full((64, 64), float(rank + 1))
    y = torch.empty_like(x)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y.copy_(x)
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dist, cpu)
    lo, hi = sharding.shard_range(args.c5_signals, rank, dist.get_world_size())
    cols = sharding.sum_over_ranks(float(hi - lo), dist, cpu)
    checksum = sharding.sum_over_ranks(float(y.sum()), dist, cpu)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "STUB (launcher test, no transform ran)", "stub": True, "value": 0.0, "unit": "Msamples/s",
                          "n_gpus": dist.get_world_size(), "gpus_requested": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / max(1, args.steps) * 1e3, "backend": args.stub_backend,
                          "c5_signals_covered": cols, "checksum_all_ranks": checksum}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    cmd = respawn_command(args, sys.argv[1:])
    if cmd is not None:
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.stub_backend:
        return stub_main(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    if world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks requested but only {torch.cuda.device_count()} HIP devices are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("WL_BENCH_FORCE_DIST") == "1":    # (the env knob exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(backend="nccl", device_id=device)       # "nccl" is RCCL on ROCm
    import wavelets_jl_amd as W
    from wavelets_jl_amd import sharding
    W._lib.load()
    W.set_kernel_path(args.path)

    label, x, wt, Ldef, dtag = make_workload(W, args.workload, device, 42 + 1000 * rank)
    L = Ldef if args.levels is None else args.levels
    # filter taps / scheme coefficients travel from rank 0 over RCCL (xGMI): the only collective
    wt = sharding.broadcast_wavelet(wt, dist, device)
    batched = args.workload == "c5"
    # the timed step is the reference's non-allocating entry point dwt!(y, x, wt, L) / dwt_oop!(y, x, scheme, L)
    # (transforms_main.jl:114-117,193-207): output array and workspace are allocated once, outside the timed region
    yout = W.similar(x)
    fn = (lambda t: W.dwtc_(yout, t, wt, L)) if batched else (lambda t: W.dwt_oop_(yout, t, wt, L))
    W.reserve_workspace(x, L)
    nsamples = x.numel()

    # device conditioning before the W warm-up steps (untimed, reported in the JSON line): clocks need about a
    # millisecond of load to ramp, and the ROCm runtime has a one-off enqueue stall the first time the host runs a few
    # hundred launches ahead -- neither belongs to the steady-state throughput this line reports
    precondition = max(0, 300 - args.warmup)
    for _ in range(precondition):
        y = fn(x)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()                      # all ranks start the warm-up (and with it the timed steps) together
    torch.cuda.synchronize()
    # the W warm-up steps run AFTER the barrier: an RCCL barrier idles the GPU for about a millisecond, long enough
    # for the clocks to drop again, and the timed steps must not start cold
    for _ in range(args.warmup):
        y = fn(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = fn(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0               # this rank's K steps, start aligned by the barrier above
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = sharding.max_over_ranks(dt, dist, device)   # the job is as slow as its slowest rank
    kernel = W.last_kernel()
    ms_per_step = dt / args.steps * 1e3
    value = world * nsamples / (dt / args.steps) / 1e6          # whole-job Msamples/s
    esize = x.element_size()
    gbps = 2 * esize * value * 1e6 / 1e9                        # algorithmic bytes: 2*N*sizeof(T) per call

    # device-only time of one step (HIP events on the launch stream), for reference
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nev = min(args.steps, 200)
    ev0.record()
    for _ in range(nev):
        y = fn(x)
    ev1.record()
    torch.cuda.synchronize()
    dev_ms_per_step = ev0.elapsed_time(ev1) / nev

    # cross-rank correctness token: sum over all ranks of each rank's coefficient sum, one 8-byte all-reduce
    # over RCCL after the timed region (single rank: its own sum)
    checksum = sharding.sum_over_ranks(float(yout.sum(dtype=torch.float64).item()), dist, device)

    out = {
        "metric": "Msamples/s, 2-D db4 dwt 8192x8192 f32" if args.workload == "c3" else "Msamples/s, " + label,
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtag, "data": "synthetic (standard normal, seed 42 + 1000*rank), resident in HBM",
        "config": {"workload": label, "L": int(L), "arrays": world, "parallelism": f"{world} independent arrays, one per GPU",
                   "kernel": kernel, "kernel_path": "generic" if args.path else "fast", "gpus_requested": args.gpus,
                   "world_size": (dist.get_world_size() if dist is not None else 1),
                   "backend": (dist.get_backend() + " (RCCL)" if dist is not None else "single process"),
                   "untimed_precondition_steps": precondition},
        "achieved_hbm_GBps_algorithmic": round(gbps, 1), "precondition_steps": precondition,
        "checksum_all_ranks": checksum,
        "device_ms_per_step": round(dev_ms_per_step, 5),
    }

    if args.workload == "c3" and not args.no_c5:
        # every rank takes part (barriers / reductions inside); the object is kept by rank 0
        del y
        c5 = c5_batched_leg(W, sharding, dist, device, rank, world, args)
        if rank == 0:
            out["c5_batched"] = c5
    if rank == 0 and world == 1 and not batched and not args.no_pipelined:
        out["pipelined"] = pipelined_leg(W, x, wt, L, args)
    if rank == 0:
        out["roofline"] = roofline_leg(W, x, wt, batched, esize, args, kernel)
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_secondary:
        del yout
        out["secondary_configs"] = secondary_leg(W, device)
    if rank == 0:
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline_leg(W, args.workload, wt, L)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST line on stdout: RCCL writes its banner through C stdio, which is fully
        # buffered on a pipe and would otherwise be flushed after Python's line at exit
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def c5_batched_leg(W, sharding, dist, device, rank, world, args):
    """BASELINE.json configs[4]: batched column-wise dwt, WT.db4, `--c5-signals` (65536) signals x 2^16 Float32, L = 16,
    the batch sharded by columns over the ranks (rank r owns sharding.shard_range(signals, r, world): 65536/N columns, no
    signal data crosses GPUs); rank 0's filter taps reach the others by one RCCL broadcast.  Strong scaling of a fixed
    batch: value = signals * 2^16 / max-over-ranks time.  Protocol as the headline: barrier + synchronize, warm-up,
    K timed steps, synchronize, MAX over ranks; checksum SUM-reduced over RCCL."""
    n = 1 << 16
    lo, hi = sharding.shard_range(args.c5_signals, rank, world)
    ncol = hi - lo
    wt = sharding.broadcast_wavelet(W.wavelet(W.WT.db4), dist, device)
    g = torch.Generator(device=device).manual_seed(4242 + 1000 * rank)
    x = torch.randn(ncol, n, generator=g, dtype=torch.float32, device=device).t()      # Julia layout: n x ncol, column = signal
    y = W.similar(x)
    fn = lambda: W.dwtc_(y, x, wt, 16)
    steps, warm = 5, 2
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt = sharding.max_over_ranks(dt, dist, device) / steps
    checksum = sharding.sum_over_ranks(float(y.sum(dtype=torch.float64).item()), dist, device)
    kernel = W.last_kernel()
    total = args.c5_signals * n
    del x, y
    W.destroy_contexts()                     # (the batch's workspace)
    torch.cuda.empty_cache()
    return {"workload": f"batched column-wise dwt db4, {args.c5_signals} signals x 2^16 f32, L=16, sharded over {world} GPU(s)",
            "signals_total": args.c5_signals, "signals_per_rank": ncol, "scaling": "strong", "steps": steps, "warmup": warm,
            "ms_per_step": round(dt * 1e3, 4), "Msamples_per_s": round(total / dt / 1e6, 1),
            "aggregate_algorithmic_GBps": round(8.0 * total / dt / 1e9, 1), "checksum_all_ranks": checksum, "kernel": kernel,
            "collectives": "1 broadcast of the taps (256 B), 1 MAX + 1 SUM all-reduce of 8 B; no signal data crosses GPUs"}


def pipelined_leg(W, x, wt, L, args, nstreams=4):
    """Reported beside `value`, never instead of it: the same K transforms issued round-robin on `nstreams` HIP streams
    (one library context and one output array per stream, same resident input).  Independent transforms -- a sequence of
    images -- overlap the latency-bound small levels of one with the bandwidth-bound first kernel of the next; `value`
    above stays the strictly sequential single-stream figure the metric is defined on."""
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [W.similar(x) for _ in range(nstreams)]
    steps = max(args.steps, 100)
    for i in range(max(steps, 400)):                  # untimed: creates the contexts, absorbs the runtime's one-off enqueue stall
        with torch.cuda.stream(streams[i % nstreams]):
            W.dwt_oop_(outs[i % nstreams], x, wt, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]):
            W.dwt_oop_(outs[i % nstreams], x, wt, L)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del outs
    W.destroy_contexts()                 # the four per-stream contexts and their workspaces
    torch.cuda.empty_cache()
    return {"streams": nstreams, "steps": steps, "ms_per_step": round(ms, 5), "value": round(x.numel() / ms / 1e3, 1), "unit": "Msamples/s",
            "achieved_hbm_GBps_algorithmic": round(2 * x.numel() * x.element_size() / ms / 1e6, 1)}


def secondary_leg(W, device):
    """Short device-timed runs of the other BASELINE.json configs (parity-test configs, not the headline):
    reported for context only."""
    res = []
    for name in ("c1", "c2", "c4", "c5"):
        label, x, wt, L, dtag = make_workload(W, name, device, 42)
        y = W.similar(x)
        fn = (lambda: W.dwtc_(y, x, wt, L)) if name == "c5" else (lambda: W.dwt_oop_(y, x, wt, L))
        W.reserve_workspace(x, L)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append({"workload": label, "L": int(L), "dtype": dtag, "ms_per_step": round(ms, 5),
                    "Msamples_per_s": round(x.numel() / ms / 1e3, 1),
                    "algorithmic_GBps": round(2 * x.numel() * x.element_size() / ms / 1e6, 1), "kernel": W.last_kernel()})
        del x, y
        torch.cuda.empty_cache()
    # the inverse of the headline config and the section 8(f) rows (3-D, modwt), same protocol
    g = torch.Generator(device="cpu").manual_seed(7)
    db4 = W.wavelet(W.WT.db4)
    extra = []
    x2 = torch.randn(8192, 8192, generator=g, dtype=torch.float32).to(device).t()
    y2 = W.similar(x2)
    extra.append(("2-D idwt db4 filter 8192x8192 f32", 13, x2, lambda: W.idwt_oop_(y2, x2, db4, 13), 2 * x2.numel() * 4))
    cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    extra.append(("2-D dwt cdf9/7 lifting 8192x8192 f32", 13, x2, lambda: W.dwt_oop_(y2, x2, cdf, 13), 2 * x2.numel() * 4))
    extra.append(("2-D idwt cdf9/7 lifting 8192x8192 f32", 13, x2, lambda: W.idwt_oop_(y2, x2, cdf, 13), 2 * x2.numel() * 4))
    batt6 = W.wavelet(W.WT.batt6)
    extra.append(("2-D dwt batt6 (59 taps) filter 8192x8192 f32", 13, x2, lambda: W.dwt_oop_(y2, x2, batt6, 13), 2 * x2.numel() * 4))
    x3 = torch.randn(512, 512, 512, generator=g, dtype=torch.float32).to(device).permute(2, 1, 0)
    y3 = W.similar(x3)
    extra.append(("3-D dwt db4 filter 512^3 f32", 9, x3, lambda: W.dwt_oop_(y3, x3, db4, 9), 2 * x3.numel() * 4))
    xm = torch.randn(1 << 24, generator=g, dtype=torch.float32).to(device)
    extra.append(("1-D modwt db4 2^24 f32 (output 2^24 x 9)", 8, xm, lambda: W.modwt(xm, db4, 8), (1 + 9) * xm.numel() * 4))
    # translation-invariant denoise (denoising.jl:36-67), default wavelet sym5, 8 x 8 spins as one device-resident batch:
    # 64 forward + 64 inverse transforms of the image per call; algorithmic bytes = (read + write) per spin and direction
    xd = torch.randn(2048, 2048, generator=g, dtype=torch.float32).to(device).t()
    extra.append(("2-D denoise TI 8x8 spins sym5 2048x2048 f32 (64 dwt + 64 idwt, fused batch)", 6, xd,
                  lambda: W.denoise(xd, TI=True), 64 * 2 * 2 * xd.numel() * 4))
    for label, L, x, fn, alg in extra:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append({"workload": label, "L": int(L), "dtype": "f32", "ms_per_step": round(ms, 5),
                    "Msamples_per_s": round(x.numel() / ms / 1e3, 1), "algorithmic_GBps": round(alg / ms / 1e6, 1),
                    "kernel": W.last_kernel()})
    del extra, x2, y2, x3, y3, xm, xd
    W.destroy_contexts()
    torch.cuda.empty_cache()
    return res


def _time_launches(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    durs = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(durs) / len(durs), durs[len(durs) // 2], durs[0]


def _time_back_to_back(fn, reps):
    """Average duration of one launch inside a train of `reps` identical launches (one event pair around the train: no host
    gaps between the launches, which is how the kernel runs inside a transform)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _rocprof_summary(kname_substr):
    """Average duration of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary of this same
    command (profiles/r02_c3_kernel_stats.csv), so that the line can be checked against profiles/ without a GPU."""
    import csv
    path = os.path.join(ROOT, "profiles", "r02_c3_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    best = None
    for r in csv.DictReader(open(path)):
        if kname_substr in r.get("Name", ""):
            if best is None or float(r["AverageNs"]) > float(best["AverageNs"]):
                best = r
    if best is None:
        return None
    return {"file": "profiles/r02_c3_kernel_stats.csv", "kernel": best["Name"], "calls": int(best["Calls"]),
            "avg_launch_ms": round(float(best["AverageNs"]) * 1e-6, 5)}


def roofline_leg(W, x, wt, batched, esize, args, main_kernel):
    """Dominant kernel = the launch that consumes the full-size input.  Its algorithmic bytes are
    2*N*sizeof(T): it reads every input sample once and writes N coefficients (8 B/sample
    f32) -- that holds for the single-level kernels and for the fused-pair kernels, which finish TWO levels
    in the same pass (3/4 N level-1 details + 1/4 N level-2 coefficients).  A call with L = 1 (L = 2
    for a fused pair) is exactly one launch of that kernel (its first-level template instance, which
    rocprofv3 --stats reports under its own name).  `frac` uses HIP events on the launch stream around a train of
    such launches (live, this run); `rocprof` repeats the computation from the committed rocprofv3 summary."""
    Ldom = 2 if main_kernel in ("k_fwd2d_stream2", "k_fwd2d_pair") else 1
    y1 = W.similar(x)
    fn1 = (lambda: W.dwtc_(y1, x, wt, Ldom)) if batched else (lambda: W.dwt_oop_(y1, x, wt, Ldom))
    reps = max(20, min(args.steps, 200))
    train_ms = _time_back_to_back(fn1, reps)
    avg_ms, med_ms, min_ms = _time_launches(fn1, reps)
    kname = W.last_kernel()
    alg_bytes = 2 * x.numel() * esize
    achieved = alg_bytes / (train_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    note = "traffic: no PMC summary committed yet"
    if os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            if j.get("kernel_short") == kname:
                traffic = j.get("hbm_bytes_per_launch")
            note = j.get("note", "")
        except Exception:
            pass
    out = {"bound": "hbm", "kernel": f"{kname} (first launch: level{'s 1-2' if Ldom == 2 else ' 1'})",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
           "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(train_ms, 5),
           "timing": f"HIP events around a train of {reps} launches on the launch stream",
           "isolated_launch_ms": {"avg": round(avg_ms, 5), "median": round(med_ms, 5), "min": round(min_ms, 5),
                                  "note": "one event pair per launch: includes the idle-stream launch latency"},
           "launches_timed": reps, "traffic_note": note,
           "frac_of_measured_copy_6290GBps": round(achieved / 6290.0, 4)}
    rp = _rocprof_summary(kname)
    if rp is not None:
        rp["achieved"] = round(alg_bytes / (rp["avg_launch_ms"] * 1e-3) / 1e9, 1)
        rp["frac"] = round(rp["achieved"] / HBM_PEAK_GBPS, 4)
        out["rocprof"] = rp
    return out


def cpu_baseline_leg(W, workload, wt, L):
    extra = {}
    """The oracle (kind 'port': literal C restatement of the reference's single-threaded loops)
    on this host's cores, on a bounded sample (about 10-30 s of CPU work)."""
    import oracle                      # the checker / baseline -- never the product path
    oracle.build()
    rng = np.random.default_rng(42)
    if workload == "c3":
        # the full 8192 x 8192 f32 array once (67.1 Msamples; about 5-20 s on one core)
        n = 8192
        xs = rng.standard_normal((n, n), dtype=np.float32)
        t0 = time.perf_counter()
        oracle.dwt_filter(xs, wt.qmf, L)
        dt = time.perf_counter() - t0
        sample = f"one full 2-D db4 dwt of the {n}x{n} f32 array, L={L}, 1 thread (the reference has no threading)"
        ns = xs.size
        # anchor against the one number the reference publishes for this path (README.md:249-250: 1-D db2 dwt of 2^20
        # Float64, 20 levels, 24.8 ms per call on unstated hardware): the same call through the oracle on this host
        x1 = rng.random(1 << 20)
        db2 = W.wavelet(W.WT.db2)
        oracle.dwt_filter(x1, db2.qmf, 20)
        t1 = time.perf_counter()
        for _ in range(5):
            oracle.dwt_filter(x1, db2.qmf, 20)
        c1_ms = (time.perf_counter() - t1) / 5 * 1e3
        extra = {"c1_anchor": {"workload": "1-D dwt db2 filter 2^20 f64, L=20, 1 thread", "oracle_ms_per_call": round(c1_ms, 2),
                               "reference_readme_ms_per_call": 24.8, "reference_hardware": "unstated (README.md:249-250)"}}
    else:
        # parity-test configs: repeat full-size (c5: a 1/64 sub-batch) transforms for about 10 s
        if workload == "c1":
            xs = rng.standard_normal(1 << 20)
            fn, what = (lambda: oracle.dwt_filter(xs, wt.qmf, L)), "full-size 1-D db2 f64 transforms"
        elif workload == "c2":
            xs = rng.standard_normal(1 << 24, dtype=np.float32)
            fn, what = (lambda: oracle.dwt_filter(xs, wt.qmf, L)), "full-size 1-D db4 f32 transforms"
        elif workload == "c4":
            xs = rng.standard_normal(1 << 24, dtype=np.float32)
            fn, what = (lambda: oracle.dwt_lifting(xs, wt, L)), "full-size 1-D cdf9/7 lifting transforms"
        else:
            xs = rng.standard_normal((1 << 16, 128), dtype=np.float32)
            fn, what = (lambda: oracle.dwtc_filter(xs, wt.qmf, L)), "transforms of 128 of the 8192 signals (1/64 sub-batch)"
        reps = 0
        t0 = time.perf_counter()
        while True:
            fn()
            reps += 1
            if time.perf_counter() - t0 > 10.0 or reps >= 200:
                break
        dt = (time.perf_counter() - t0) / reps
        sample = f"{reps} {what}, 1 thread"
        ns = xs.size
    out = {"value": round(ns / dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": sample, "seconds": round(dt, 2), "host_cores_available": os.cpu_count()}
    out.update(extra)
    return out


if __name__ == "__main__":
    main()
