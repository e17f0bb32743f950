#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DWT backend (driver contract: ONE JSON line, the last line on stdout).

A "step" is one full forward transform through the non-allocating entry point (dwt!(y, x, wt, L) / dwtc) of synthetic
data already resident in HBM; output array and workspace are allocated once, outside the timed region.

  --gpus 1 (default)   workload C3 = BASELINE.json configs[2], the configuration the metric is quoted on:
                       2-D dwt, WT.db4 filter bank, 8192 x 8192 Float32, L = 13 (API default).  The K timed steps rotate over
                       THREE distinct input arrays (768 MiB together, beyond the 256 MiB Infinity Cache), so no step finds its
                       input cached by the previous one.  `by_depth` repeats the measurement at L = 1, 4, 13.
  --gpus N > 1         workload C5 = configs[4]: batched column-wise dwt, WT.db4, 65536 signals x 2^16 Float32, L = 16, the
                       batch block-partitioned over the N ranks (one process per GPU, sharding.shard_range; rank 0's taps reach
                       the other ranks by one RCCL broadcast; no signal data crosses GPUs).  value = whole batch / max-over-ranks
                       time ("scaling": "strong").  Nested: `c3_weak_scaling` (every rank its own 8192 x 8192 image) and
                       `single_gpu_same_batch` (the whole batch on rank 0 alone, measured in the same job: the N = 1 point).
  --gpus 1 --workload c5   the same line for N = 1 (the object `c5_batched` of the default line has the same keys).
                       Started plainly (`python bench.py --gpus N`) the script re-launches ITSELF through torch.distributed.run
                       with N ranks; under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.

Objects on the JSON line:
  roofline       the dominant kernel = the launch that consumes the full-size input (C3: k_fwd2d_pair, levels 1-2 fused; C5: every
                 rank's first k_fwd1d_multi pass, levels 1-4): algorithmic bytes / average launch duration, HIP events on the launch
                 stream around a train of such launches; `rocprof` = the same figure from the committed rocprofv3 summary under
                 profiles/; `traffic` = HBM bytes per launch from the PMC counters, MEASURED IN THIS RUN for C3 when rocprofv3 is
                 present (two separate --pmc passes of tools/wlbench.bin), else the committed collection
  by_depth       ms per transform and fraction of the 8 TB/s roofline at L = 1, 4, 13
  cpu_baseline   the oracle's sources (literal C restatement of the reference's single-threaded loops) built -O3 -march=native
                 on THIS host and timed on a bounded sample: 1 warm-up + 3 repetitions, median; 1 core (the reference has no
                 threading) and, beside it, the OpenMP-over-lines variant on all host cores
  secondary_configs   the other BASELINE configs and section-8(f) rows, cache-cold (5 inputs in rotation for the 64 MiB 1-D configs,
                 3 for the 256 MiB 2-D ones): >= 20 repetitions, one HIP event pair per repetition, median and minimum; C2 and C4
                 carry their own `roofline` object
  reference_gpu_benchmark_shapes   the shapes of the reference's benchmark/gpu_benchmark.jl through the allocating calls

Other workloads (parity-test configs, not the headline): --workload c1|c2|c4|c5.
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_ROUND = "r06"      # profiles/<round>_<config>_kernel_stats.csv are the committed rocprofv3 summaries of these commands
NROT = 3                   # distinct input arrays the timed steps rotate over (8192^2: 3 x 256 MiB > the 256 MiB Infinity Cache)
NROT_1D = 5                # ... for the 64 MiB 1-D configs (5 x 64 MiB in + 64 MiB out > 256 MiB)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--workload", default=None, choices=["c1", "c2", "c3", "c4", "c5"],
                    help="default: c3 on one GPU, the sharded c5 batch on several")
    ap.add_argument("--levels", type=int, default=None, help="override L (default: maxtransformlevels)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--path", type=int, default=0, help="0 fast kernels, 1 generic kernels only")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other BASELINE configs")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 batch object of the one-GPU line")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the 4-stream leg (profiling runs: overlapping launches skew per-kernel statistics)")
    ap.add_argument("--no-depths", action="store_true", help="skip the by_depth object")
    ap.add_argument("--c5-signals", type=int, default=65536, help="total signals of the C5 batch (BASELINE: 65536)")
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


# ---------------------------------------------------------------------------------------------------------------------
# host placement: one rank process per GPU; the host leg of a step (a few ctypes calls, ~30x longer than nothing but far shorter than the
# 7 ms C5 shard) should not migrate between sockets -- pin the rank to the cores next to its GPU when sysfs says which they are
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/bus/pci/devices/*/local_cpulist)."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def local_cpulist_file(domain, bus, dev, sysfs="/sys"):
    return os.path.join(sysfs, "bus", "pci", "devices", f"{domain:04x}:{bus:02x}:{dev:02x}.0", "local_cpulist")


def pin_rank_to_gpu_cores(pci, sysfs="/sys", apply=True):
    """pci = (domain, bus, device) of this rank's GPU.  Returns what was done, for the JSON line."""
    info = {"pinned": False}
    try:
        path = local_cpulist_file(pci[0], pci[1], pci[2], sysfs)
        with open(path) as f:
            cpus = parse_cpulist(f.read())
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cpus
        cpus = [c for c in cpus if c in allowed]
        info["source"] = path
        if cpus and len(cpus) < len(allowed):
            if apply:
                os.sched_setaffinity(0, cpus)
            info.update({"pinned": True, "cpus": len(cpus), "first": cpus[0], "last": cpus[-1]})
        else:
            info["reason"] = "local_cpulist does not narrow the allowed set"
    except (OSError, ValueError, AttributeError, IndexError) as e:
        info["reason"] = type(e).__name__
    return info


# ---------------------------------------------------------------------------------------------------------------------
# timing helpers (device side: HIP events on the launch stream = torch's current stream, which is the stream the library
# launches on)
def _event_train_ms(fns, reps, chunk=10):
    """ms per call inside a back-to-back train of `reps` calls (fns are used round-robin).  The train is enqueued without a
    gap but timed in chunks of `chunk` calls (events between the chunks); the figure is the MEDIAN chunk, so that a single
    host hiccup (a garbage collection, an interrupt) inside a 100-call train cannot triple the figure."""
    for i in range(min(5, reps)):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    nchunks = max(1, reps // chunk)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nchunks + 1)]
    evs[0].record()
    k = 0
    for c in range(nchunks):
        for _ in range(chunk if c < nchunks - 1 else reps - chunk * (nchunks - 1)):
            fns[k % len(fns)]()
            k += 1
        evs[c + 1].record()
    torch.cuda.synchronize()
    per = []
    for c in range(nchunks):
        ncall = chunk if c < nchunks - 1 else reps - chunk * (nchunks - 1)
        per.append(evs[c].elapsed_time(evs[c + 1]) / ncall)
    return statistics.median(per)


def _event_each_ms(fns, reps, warm=5, warm_ms=40.0):
    """one event pair per call, calls enqueued back to back: (median, min, mean) ms.  Untimed conditioning first: `warm` calls, then
    as many more as fill `warm_ms` of device time (at most 400) -- after the host-side set-up of a leg the device has been idle
    for seconds, and a handful of sub-millisecond calls does not bring its clocks back (8192^2 Float64: 0.32-0.36 ms after five
    calls, 0.28 ms in steady state)."""
    for i in range(warm):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(3):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    est = max(a.elapsed_time(b) / 3.0, 1e-3)
    for i in range(min(400, max(0, int(warm_ms / est) - 3))):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(evs):
        a.record()
        fns[i % len(fns)]()
        b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in evs)
    return statistics.median(d), d[0], sum(d) / len(d)


def make_workload(W, name, device, seed, ncols=8192):
    """returns (label, x (device tensor), wt, default L, dtype tag)"""
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
        # a shard of the 65536 x 2^16 batch: `ncols` signals of length 2^16 (8192 = the per-GPU shard on 8 GPUs), generated on the device
        gd = torch.Generator(device=device).manual_seed(seed)
        x = torch.randn(ncols, 1 << 16, generator=gd, dtype=torch.float32, device=device).t()
        return f"batched column-wise dwt db4, {ncols} signals x 2^16 f32", x, W.wavelet(WT.db4), 16, "f32"
    raise ValueError(name)


# ---------------------------------------------------------------------------------------------------------------------
class StubBackend:
    """CPU stand-in for the transform calls (tests/test_bench_launch.py, gloo ranks): the launcher, rendezvous, shard partition,
    reductions and the assembly of the multi-GPU JSON line are what is under test; nothing is measured and the line says so."""
    name = "stub"

    def __init__(self, device, rank):
        self.device, self.rank = device, rank

    def make_shard(self, ncol, seed):
        x = torch.full((64, 8), float(self.rank + 1))         # rank r's shard holds r + 1 (the checksum test relies on it)
        return x, torch.empty_like(x)

    def make_image(self, seed):
        x = torch.full((64, 64), float(self.rank + 1))
        return x, torch.empty_like(x)

    def dwtc(self, y, x):
        y.copy_(x)

    def dwt2(self, y, x):
        y.copy_(x)

    def sync(self):
        pass

    def kernel(self):
        return "stub-copy"


class HipBackend:
    name = "hip"

    def __init__(self, W, sharding, dist, device):
        self.W, self.device = W, device
        self.db4 = sharding.broadcast_wavelet(W.wavelet(W.WT.db4), dist, device)     # rank 0's taps, over RCCL

    def make_shard(self, ncol, seed):
        _, x, _, _, _ = make_workload(self.W, "c5", self.device, seed, ncols=ncol)
        y = self.W.similar(x)
        self.W.reserve_workspace(x, 16)
        return x, y

    def make_image(self, seed):
        _, x, _, _, _ = make_workload(self.W, "c3", self.device, seed)
        return x, self.W.similar(x)

    def dwtc(self, y, x):
        self.W.dwtc_(y, x, self.db4, 16)

    def dwtc_levels(self, y, x, L):
        self.W.dwtc_(y, x, self.db4, L)

    def dwt2(self, y, x):
        self.W.dwt_oop_(y, x, self.db4, 13)

    def sync(self):
        torch.cuda.synchronize()

    def kernel(self):
        return self.W.last_kernel()


def timed_steps(step, backend, dist, sharding, device, steps, warmup, precondition=0):
    """The driver's protocol: [untimed conditioning] barrier + synchronize, W warm-up steps, synchronize, K timed steps,
    synchronize, barrier; returns the MAX over ranks of this rank's wall time for the K steps (seconds).
    Python's cyclic garbage collector is off inside (as in `timeit`): with torch imported, the first full collection of a
    process takes 35-45 ms of host time -- it arrives around the 1240th library call (70 000 container allocations) and, landing
    in a timed region whose calls are shorter than the host's lead, shows up as a stall of the whole queue (round 2 blamed the
    ROCm runtime for it; tools/exp measured the same stall at the same call index for 3- and 6-launch calls, and none after
    gc.freeze())."""
    gc.collect()
    gc.freeze()
    gc.disable()
    try:
        return _timed_steps(step, backend, dist, sharding, device, steps, warmup, precondition)
    finally:
        gc.enable()


def _timed_steps(step, backend, dist, sharding, device, steps, warmup, precondition=0):
    for i in range(precondition):
        step(i)
    backend.sync()
    if dist is not None:
        dist.barrier()                      # all ranks start the warm-up (and with it the timed steps) together
    backend.sync()
    # the W warm-up steps run AFTER the barrier: an RCCL barrier idles the GPU for about a millisecond, long enough for the
    # clocks to drop again, and the timed steps must not start cold
    for i in range(warmup):
        step(i)
    backend.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    backend.sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    backend.sync()
    return sharding.max_over_ranks(dt, dist, device)


def rccl_info(dist, device, backend_name):
    """What the collective library itself reports: its version, and the number of ranks as counted by an all-reduce of ones
    (not torch's bookkeeping)."""
    if dist is None:
        return {"backend": "single process", "ranks_counted_by_allreduce": 1}
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    info = {"backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""),
            "ranks_counted_by_allreduce": int(one.item()), "world_size": dist.get_world_size()}
    if backend_name == "hip":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                       # pragma: no cover
            info["rccl_version"] = f"unavailable ({type(e).__name__})"
    return info


def c5_roofline(backend, x, y, ncol, rank, dist, sharding, device):
    """`roofline` of the C5 line: the first pass of every rank's shard -- k_fwd1d_multi, levels 1-4 in one launch (an L = 4 call is
    exactly that launch): it reads the shard once and writes it once = 8 B/sample.  Per-rank figure (HIP events on the launch
    stream around a back-to-back train, median of chunks); the line carries rank 0's and the slowest rank's duration."""
    if backend.name == "stub":
        return {"bound": "hbm", "kernel": "stub-copy", "achieved": 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 0.0, "traffic": None,
                "note": "stub run: not measured"}
    ms = _event_train_ms([lambda: backend.dwtc_levels(y, x, 4)], 20, chunk=5)
    kname = backend.kernel()
    ms_max = sharding.max_over_ranks(ms, dist, device)
    alg = 2 * ncol * (1 << 16) * 4
    ach = alg / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": f"{kname} (first launch of every shard: levels 1-4)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "algorithmic_bytes_per_launch": alg,
           "avg_launch_ms": round(ms, 5), "avg_launch_ms_slowest_rank": round(ms_max, 5), "per": "rank (one GPU)", "signals_in_launch": ncol,
           "timing": "HIP events on the launch stream inside a back-to-back train of 20 launches (median of chunks of 5), rank 0",
           "traffic_note": "traffic: see profiles/pmc_c5.json (8192-signal shard, static)"}
    pmc = os.path.join(ROOT, "profiles", "pmc_c5.json")
    if os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            if j.get("kernel_short") == kname and j.get("signals_in_launch"):
                out["traffic"] = int(j["hbm_bytes_per_launch"] * (ncol / j["signals_in_launch"]))
                out["traffic_note"] = j.get("note", "") + f" (static: profiles/pmc_c5.json, scaled from {j['signals_in_launch']} to {ncol} signals)"
        except Exception:
            pass
    rp = _rocprof_summary(kname, "c5")
    if rp is not None:
        # the committed summary is `bench.py --workload c5` on one GPU: the whole batch in slab launches of 32768 signals
        rp["signals_per_launch"] = 32768
        rp["note"] = "committed summary: the whole 65536-signal batch on one GPU, two slab launches of 32768 signals per transform"
        rp["achieved"] = round(2 * 32768 * (1 << 16) * 4 / (rp["avg_launch_ms"] * 1e-3) / 1e9, 1)
        rp["frac"] = round(rp["achieved"] / HBM_PEAK_GBPS, 4)
        out["rocprof"] = rp
    return out


def multi_gpu_line(args, rank, world, dist, sharding, device, backend, steps=None, warm=None, nested=True):
    """The sharded C5 batch as the top-level metric: `--gpus N > 1`, `--gpus 1 --workload c5`, and (with few steps) the
    `c5_batched` object of the one-GPU C3 line -- the same keys everywhere, so the N = 1 point and the N > 1 points of the
    scaling curve compare like for like.  nested: also report the C3 weak-scaling figure (every rank its own image) and, for
    N > 1, the whole batch on rank 0 alone measured in the SAME job."""
    n = 1 << 16
    steps = args.steps if steps is None else steps
    lo, hi = sharding.shard_range(args.c5_signals, rank, world)
    ncol = hi - lo
    x, y = backend.make_shard(ncol, 4242 + 1000 * rank)
    warm = min(args.warmup, 20) if warm is None else warm   # a 16 GiB / N shard per step: a handful of warm-up steps reach steady clocks
    dt = timed_steps(lambda i: backend.dwtc(y, x), backend, dist, sharding, device, steps, warm)
    kernel = backend.kernel()
    checksum = sharding.sum_over_ranks(float(y.double().sum().item()), dist, device)
    cols = sharding.sum_over_ranks(float(ncol), dist, device)
    total = args.c5_signals * n
    ms = dt / steps * 1e3
    value = total / (dt / steps) / 1e6
    roof = c5_roofline(backend, x, y, ncol, rank, dist, sharding, device)
    del x, y
    if backend.name == "hip":
        backend.W.destroy_contexts()
        torch.cuda.empty_cache()
    out = {
        "metric": "Msamples/s, batched column-wise db4 dwt 65536 x 2^16 f32 sharded over the GPUs (BASELINE.json configs[4])",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": steps, "warmup": warm,
        "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (standard normal, generated on each device, seed 4242 + 1000*rank), resident in HBM",
        "config": {"workload": f"batched column-wise dwt db4 {args.c5_signals} x 2^16 f32, L=16, {world} shards",
                   "signals_total": args.c5_signals, "signals_per_rank": ncol, "signals_covered_all_ranks": cols,
                   "parallelism": f"column block partition over {world} ranks, one process per GPU, no data-path collective",
                   "collectives": "1 broadcast of the taps (256 B), MAX / SUM all-reduces of 8 B; no signal data crosses GPUs",
                   "kernel": kernel, "gpus_requested": args.gpus, "warmup_requested": args.warmup},
        "gpus_requested": args.gpus,
        "achieved_hbm_GBps_algorithmic": round(8.0 * total / (dt / steps) / 1e9, 1),
        "hbm_frac_whole_transform_per_gpu": round(8.0 * total / (dt / steps) / 1e9 / world / HBM_PEAK_GBPS, 4),
        "checksum_all_ranks": checksum, "c5_signals_covered": cols,
        "roofline": roof,
        "rccl": rccl_info(dist, device, backend.name),
    }
    if nested and world > 1 and backend.name == "hip":
        # the N = 1 point of the same curve, measured in this job: rank 0 alone transforms the WHOLE batch (16 GiB in, 16 GiB out,
        # 16 GiB workspace) while the other ranks wait at the barrier
        single = None
        if rank == 0:
            try:
                xa, ya = backend.make_shard(args.c5_signals, 4242)
                med, mn, _ = _event_each_ms([lambda: backend.dwtc(ya, xa)], 6, warm=2, warm_ms=0.0)
                single = {"workload": "the whole batch on rank 0 alone", "ms_per_step_median": round(med, 4), "ms_per_step_min": round(mn, 4),
                          "Msamples_per_s": round(total / med / 1e3, 1), "speedup_of_this_line": round(med / ms, 3)}
                del xa, ya
                backend.W.destroy_contexts()
                torch.cuda.empty_cache()
            except Exception as e:                        # pragma: no cover  (e.g. not enough free HBM on rank 0)
                single = {"error": type(e).__name__}
        if dist is not None:
            dist.barrier()
        out["single_gpu_same_batch"] = single
    if nested:
        # every rank its own 8192 x 8192 image, rotating over NROT inputs (weak scaling of independent images)
        imgs = [backend.make_image(42 + 1000 * rank + 17 * j) for j in range(NROT)]
        yout = imgs[0][1]
        steps3 = max(20, min(steps * 10, 200))
        dt3 = timed_steps(lambda i: backend.dwt2(yout, imgs[i % NROT][0]), backend, dist, sharding, device, steps3, min(args.warmup * 10, 100),
                          precondition=100 if backend.name == "hip" else 0)
        k3 = backend.kernel()
        ms3 = dt3 / steps3 * 1e3
        nimg = imgs[0][0].numel()
        out["c3_weak_scaling"] = {"workload": "2-D dwt db4 filter 8192x8192 f32, L=13, one independent image per GPU", "scaling": "weak",
                                  "steps": steps3, "ms_per_step": round(ms3, 5), "value": round(world * nimg / ms3 / 1e3, 1),
                                  "unit": "Msamples/s", "kernel": k3, "inputs_rotated": NROT}
    if backend.name == "stub":
        out["stub"] = True
        out["metric"] = "STUB (launcher test, no transform ran): " + out["metric"]
    return out


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
    force_dist = os.environ.get("WL_BENCH_FORCE_DIST") == "1"     # (exercises the RCCL path on one GPU)
    if args.stub_backend:
        import torch.distributed as dist
        from wavelets_jl_amd import sharding
        dist.init_process_group(backend=args.stub_backend)
        device = torch.device("cpu")
        out = multi_gpu_line(args, rank, world, dist, sharding, device, StubBackend(device, rank))
        if rank == 0:
            out["cpu_baseline"] = {"value": 0.0, "unit": "Msamples/s", "cores": 1, "kind": "port", "sample": "stub run: not measured"}
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    if world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks requested but only {torch.cuda.device_count()} HIP devices are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    affinity = None
    if world > 1:
        pr = torch.cuda.get_device_properties(local_rank)
        affinity = pin_rank_to_gpu_cores((getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", 0)))
    dist = None
    if world > 1 or force_dist:
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
    # no cyclic garbage collection during the measurements (see timed_steps): reference counting still frees everything the
    # bench allocates; the legs below time calls of 20-200 us, a full collection with torch loaded costs 40 ms
    gc.collect()
    gc.freeze()
    gc.disable()

    if (world > 1 and args.workload is None) or args.workload == "c5":
        # the sharded C5 batch: the metric of every N > 1 line, and of `--gpus 1 --workload c5` (the like-for-like N = 1 point)
        out = multi_gpu_line(args, rank, world, dist, sharding, device, HipBackend(W, sharding, dist, device),
                             steps=(args.steps if world > 1 else min(args.steps, 60)), nested=(args.workload is None))
        if rank == 0 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline_leg(W, "c5", W.wavelet(W.WT.db4), 16)
        if affinity is not None:
            out["rank0_cpu_affinity"] = affinity
        finish(out, rank, dist)
        return

    workload = args.workload or "c3"
    label, x, wt, Ldef, dtag = make_workload(W, workload, device, 42 + 1000 * rank)
    L = Ldef if args.levels is None else args.levels
    # filter taps / scheme coefficients travel from rank 0 over RCCL (xGMI): the only collective
    wt = sharding.broadcast_wavelet(wt, dist, device)
    batched = workload == "c5"
    # C3 rotates over NROT distinct inputs: 3 x 256 MiB do not fit the 256 MiB Infinity Cache, so no step can find its input
    # (or the previous step's) cached
    xs = [x]
    if workload == "c3":
        for j in range(1, NROT):
            xs.append(make_workload(W, workload, device, 42 + 1000 * rank + 17 * j)[1])
    elif workload in ("c1", "c2", "c4"):
        # the 1-D configs rotate too (round 5): 5 x 64 MiB in + 64 MiB out do not fit the Infinity Cache, so `--workload c2 / c4` is as
        # cache-cold as the secondary legs of the C3 line (C1: 5 x 8 MiB stay cached whatever one does -- stated in the line)
        for j in range(1, NROT_1D):
            xs.append(make_workload(W, workload, device, 42 + 1000 * rank + 17 * j)[1])
    # the timed step is the reference's non-allocating entry point dwt!(y, x, wt, L) / dwt_oop!(y, x, scheme, L)
    # (transforms_main.jl:114-117,193-207): output array and workspace are allocated once, outside the timed region
    yout = W.similar(x)
    fn = (lambda t: W.dwtc_(yout, t, wt, L)) if batched else (lambda t: W.dwt_oop_(yout, t, wt, L))
    W.reserve_workspace(x, L)
    nsamples = x.numel()
    esize = x.element_size()

    class _B:                                   # (timed_steps only needs sync())
        @staticmethod
        def sync():
            torch.cuda.synchronize()
    # The other end of the range first (round 5): whole calls issued into an IDLE device -- the very first call of the process
    # (code-object load, first-touch of the workspace), then single calls each after 50 ms of idleness (clocks down, caches
    # cold), one HIP event pair per call, no conditioning of any kind.
    isolated = None
    if rank == 0 and workload == "c3":
        isolated = isolated_call_leg(fn, xs)
    # device conditioning before the W warm-up steps (untimed, reported in the JSON line): clocks need about a millisecond of
    # load to ramp before the steady-state throughput this line reports
    precondition = max(0, 300 - args.warmup)
    dt = timed_steps(lambda i: fn(xs[i % len(xs)]), _B, dist, sharding, device, args.steps, args.warmup, precondition)
    kernel = W.last_kernel()
    ms_per_step = dt / args.steps * 1e3
    value = world * nsamples / (dt / args.steps) / 1e6          # whole-job Msamples/s
    gbps = 2 * esize * value * 1e6 / 1e9                        # algorithmic bytes: 2*N*sizeof(T) per call

    # device-only time of one step (HIP events on the launch stream), for reference
    dev_ms_per_step = _event_train_ms([(lambda t=t: fn(t)) for t in xs], min(args.steps, 200))

    # cross-rank correctness token: sum over all ranks of each rank's coefficient sum, one 8-byte all-reduce
    checksum = sharding.sum_over_ranks(float(yout.sum(dtype=torch.float64).item()), dist, device)

    out = {
        "metric": "Msamples/s, 2-D db4 dwt 8192x8192 f32" if workload == "c3" else "Msamples/s, " + label,
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtag, "data": "synthetic (standard normal, seed 42 + 1000*rank + 17*j), resident in HBM",
        "config": {"workload": label, "L": int(L), "arrays": world, "parallelism": f"{world} independent arrays, one per GPU",
                   "inputs_rotated": len(xs), "kernel": kernel, "kernel_path": "generic" if args.path else "fast",
                   "gpus_requested": args.gpus, "untimed_precondition_steps": precondition},
        "gpus_requested": args.gpus,
        "achieved_hbm_GBps_algorithmic": round(gbps, 1), "hbm_frac_whole_transform": round(gbps / HBM_PEAK_GBPS, 4),
        "precondition_steps": precondition, "untimed_steps_before_timing": args.warmup + precondition, "checksum_all_ranks": checksum,
        "device_ms_per_step": round(dev_ms_per_step, 5),
        "rccl": rccl_info(dist, device, "hip"),
    }
    if isolated is not None:
        out["isolated_call_ms"] = isolated
    if rank == 0 and workload == "c3" and not args.no_depths:
        out["by_depth"] = by_depth_leg(W, xs, yout, wt, esize)
    if rank == 0 and world == 1 and workload == "c3" and not batched and os.path.exists(W._lib.LIB_PATHS["fused"]):
        out["fused_mode"] = fused_mode_leg(W, xs, yout, wt, L, esize)
    if rank == 0:
        out["roofline"] = roofline_leg(W, xs, wt, batched, esize, max(20, min(args.steps, 200)), kernel, tag=workload,
                                       live_pmc=(world == 1 and workload == "c3"))
    if rank == 0 and world == 1 and not batched and not args.no_pipelined:
        out["pipelined"] = pipelined_leg(W, xs, wt, L, args)
    if workload == "c3" and world == 1 and not args.no_c5:
        del yout, xs, x
        torch.cuda.empty_cache()
        out["c5_batched"] = multi_gpu_line(args, rank, world, dist, sharding, device, HipBackend(W, sharding, dist, device),
                                           steps=12, warm=6, nested=False)
        out["c5_batched"]["note"] = ("the N = 1 point of the multi-GPU curve: the same object `--gpus N` prints as its top level "
                                     "(12 steps after 6 warm-up steps here; `--gpus 1 --workload c5` runs it with --steps)")
    if rank == 0 and world == 1 and workload == "c3" and not args.no_secondary:
        xs = x = yout = None
        torch.cuda.empty_cache()
        out["secondary_configs"] = secondary_leg(W, device)
        out["reference_gpu_benchmark_shapes"] = reference_shapes_leg(W, device)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_leg(W, workload, wt, L)
    finish(out, rank, dist)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def slim_line(out, limit=6000):
    """The LAST stdout line, short enough for a reader that keeps only a few KB of the tail (round-5 review: the 25 KB line lost
    `by_depth`, `fused_mode`, `c5_batched` and the C1/C2/C4 rows): the contract keys and every measured number, without the prose.
    The complete objects (protocol notes, every secondary row's fields) are printed on the line BEFORE it and written to
    bench_full.json."""
    keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "gpus_requested", "achieved_hbm_GBps_algorithmic", "hbm_frac_whole_transform",
            "hbm_frac_whole_transform_per_gpu", "untimed_steps_before_timing", "checksum_all_ranks", "device_ms_per_step", "rccl",
            "pipelined", "c5_signals_covered"]
    s = _pick(out, keep)
    if "config" in s:
        s["config"] = {k: v for k, v in s["config"].items() if k not in ("parallelism", "collectives")}
    if "roofline" in out:
        r = out["roofline"]
        s["roofline"] = _pick(r, ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                  "avg_launch_ms", "launches_timed", "frac_of_measured_copy_6290GBps"])
        if isinstance(r.get("isolated_launch_ms"), dict):
            s["roofline"]["isolated_launch_ms"] = _pick(r["isolated_launch_ms"], ["median", "min"])
        if isinstance(r.get("rocprof"), dict):
            s["roofline"]["rocprof"] = _pick(r["rocprof"], ["file", "calls", "avg_launch_ms", "frac"])
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        s["cpu_baseline"] = _pick(c, ["value", "unit", "cores", "kind", "seconds", "host_cores_available"])
        s["cpu_baseline"]["sample"] = str(c.get("sample", ""))[:150]
        if isinstance(c.get("all_cores"), dict):
            s["cpu_baseline"]["all_cores"] = _pick(c["all_cores"], ["value", "cores", "seconds"])
    if "by_depth" in out:
        s["by_depth"] = {k: _pick(v, ["ms_per_step", "frac", "launches"]) for k, v in out["by_depth"].items()}
    if "isolated_call_ms" in out:
        s["isolated_call_ms"] = _pick(out["isolated_call_ms"], ["first_call_of_the_process", "median_of_idle_started_calls", "min", "max", "calls"])
    if "fused_mode" in out:
        s["fused_mode"] = _pick(out["fused_mode"], ["library", "ms_per_step", "value", "hbm_frac_whole_transform", "inverse_ms_per_step",
                                                     "exact_same_protocol", "gain"])
    if "c5_batched" in out:
        c5 = out["c5_batched"]
        s["c5_batched"] = _pick(c5, ["value", "unit", "ms_per_step", "steps", "warmup", "scaling", "hbm_frac_whole_transform_per_gpu"])
        if isinstance(c5.get("roofline"), dict):
            s["c5_batched"]["roofline"] = _pick(c5["roofline"], ["kernel", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"])
    for nested in ("c3_weak_scaling", "single_gpu_same_batch"):
        if isinstance(out.get(nested), dict):
            s[nested] = _pick(out[nested], ["value", "unit", "ms_per_step", "n_gpus", "scaling", "hbm_frac_whole_transform",
                                            "hbm_frac_whole_transform_per_gpu", "note"])
    if "secondary_configs" in out:
        # one short row per secondary configuration: [workload, L, ms_per_step, fraction of 8 TB/s (algorithmic bytes), kernel, bound]
        rows = []
        for r in out["secondary_configs"]:
            rf = r.get("roofline") if isinstance(r.get("roofline"), dict) else {}
            rows.append([r.get("workload"), r.get("L"), r.get("ms_per_step"), r.get("frac"), r.get("kernel"), rf.get("bound", "hbm"), rf.get("frac")])
        s["secondary_configs"] = {"columns": ["workload", "L", "ms_per_step", "hbm_frac_algorithmic", "kernel", "bound", "frac_of_bound"], "rows": rows}
    s["full_line"] = "the line before this one, and bench_full.json"
    text = json.dumps(s, separators=(",", ":"))
    # still too long (it should not be): drop the least important tables first
    for k in ("pipelined", "rccl", "secondary_configs", "c5_batched", "fused_mode"):
        if len(text) <= limit:
            break
        s.pop(k, None)
        text = json.dumps(s, separators=(",", ":"))
    return text


def finish(out, rank, dist):
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
        full = json.dumps(out)
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_full.json"), "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
        print(full, flush=True)
        print(slim_line(out), flush=True)


def isolated_call_leg(fn, xs, n=7, idle_s=0.05):
    """Whole-call device time at the COLD end: call 0 is the first call the process makes (module load, untouched workspace); calls
    1..n-1 each start after `idle_s` of idleness.  One event pair per call on the launch stream.  `ms_per_step` (steady state, after
    the reported conditioning) and this object bracket what a caller sees."""
    times = []
    for i in range(n):
        torch.cuda.synchronize()
        time.sleep(idle_s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(xs[i % len(xs)])
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    rest = sorted(times[1:])
    return {"first_call_of_the_process": round(times[0], 4), "median_of_idle_started_calls": round(rest[len(rest) // 2], 4),
            "min": round(rest[0], 4), "max": round(rest[-1], 4), "calls": n - 1, "idle_before_each_call_s": idle_s,
            "protocol": "one HIP event pair around ONE whole transform call issued into an idle device; no warm-up, no conditioning; "
                        "inputs rotate"}


def fused_mode_leg(W, xs, yout, wt, L, esize):
    """The opt-in fused arithmetic mode (libwavelets_mi355x_fma.so: the same kernels with FMA contraction; agrees with the reference to
    1e-6 sqrt(L) relative L2, tests/test_gpu_fused.py) on the headline workload, same protocol as `device_ms_per_step`.  Reported beside
    `value`, never instead of it: the headline is the bit-exact mode."""
    res = {"mode": "fused (FMA contraction allowed; tolerance-level agreement, not bit-exact)", "library": os.path.basename(W._lib.LIB_PATHS["fused"])}
    try:
        W.set_arithmetic("fused")
        W.reserve_workspace(xs[0], L)
        for i in range(100):
            W.dwt_oop_(yout, xs[i % len(xs)], wt, L)
        ms = _event_train_ms([(lambda t=t: W.dwt_oop_(yout, t, wt, L)) for t in xs], 100)
        alg = 2 * xs[0].numel() * esize
        res.update({"ms_per_step": round(ms, 5), "value": round(xs[0].numel() / ms / 1e3, 1), "unit": "Msamples/s",
                    "hbm_frac_whole_transform": round(alg / ms / 1e6 / HBM_PEAK_GBPS, 4), "kernel": W.last_kernel()})
        msi = _event_train_ms([(lambda t=t: W.idwt_oop_(yout, t, wt, L)) for t in xs], 100)
        res["inverse_ms_per_step"] = round(msi, 5)
    finally:
        W.set_arithmetic("exact")
        W.reserve_workspace(xs[0], L)
    for i in range(100):                                   # the same conditioning as the fused leg (a fresh context after the switch)
        W.dwt_oop_(yout, xs[i % len(xs)], wt, L)
    mse = _event_train_ms([(lambda t=t: W.dwt_oop_(yout, t, wt, L)) for t in xs], 100)
    msie = _event_train_ms([(lambda t=t: W.idwt_oop_(yout, t, wt, L)) for t in xs], 100)
    res["exact_same_protocol"] = {"ms_per_step": round(mse, 5), "inverse_ms_per_step": round(msie, 5)}
    res["gain"] = {"forward": round(mse / res["ms_per_step"], 4), "inverse": round(msie / res["inverse_ms_per_step"], 4)} if "ms_per_step" in res else None
    return res


def by_depth_leg(W, xs, yout, wt, esize):
    """the C3 transform at L = 1, 4 and 13 -- ms per transform (back-to-back train of 100 calls rotating
    over the inputs, HIP events every 10 calls, median chunk) and the fraction of the 8 TB/s roofline for the algorithmic 2*N*sizeof(T) bytes, which do not
    depend on L."""
    res = {}
    alg = 2 * xs[0].numel() * esize
    for Ld in (1, 4, 13):
        ms = _event_train_ms([(lambda t=t: W.dwt_oop_(yout, t, wt, Ld)) for t in xs], 100)
        res[f"L={Ld}"] = {"ms_per_step": round(ms, 5), "Msamples_per_s": round(xs[0].numel() / ms / 1e3, 1),
                          "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac": round(alg / ms / 1e6 / HBM_PEAK_GBPS, 4),
                          "launches": W.last_kernel()}
    return res


def pipelined_leg(W, xs, wt, L, args, nstreams=4):
    """Reported beside `value`, never instead of it: the same K transforms issued round-robin on `nstreams` HIP streams
    (one library context and one output array per stream).  Independent transforms -- a sequence of images -- overlap the
    latency-bound small levels of one with the bandwidth-bound first kernel of the next; `value` above stays the strictly
    sequential single-stream figure the metric is defined on."""
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [W.similar(xs[0]) for _ in range(nstreams)]
    steps = max(args.steps, 100)
    for i in range(max(steps, 400)):                  # untimed: creates the per-stream contexts, ramps the clocks
        with torch.cuda.stream(streams[i % nstreams]):
            W.dwt_oop_(outs[i % nstreams], xs[i % len(xs)], wt, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]):
            W.dwt_oop_(outs[i % nstreams], xs[i % len(xs)], wt, L)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del outs
    W.destroy_contexts()                 # the per-stream contexts and their workspaces
    torch.cuda.empty_cache()
    x = xs[0]
    return {"streams": nstreams, "steps": steps, "ms_per_step": round(ms, 5), "value": round(x.numel() / ms / 1e3, 1), "unit": "Msamples/s",
            "achieved_hbm_GBps_algorithmic": round(2 * x.numel() * x.element_size() / ms / 1e6, 1)}


def secondary_leg(W, device, reps=20):
    """Device-timed runs of the other BASELINE.json configs and of the section-8(f) rows (parity-test configs, not the
    headline).  Protocol: untimed conditioning calls (5, then enough to fill 40 ms of device time), then `reps` calls enqueued
    back to back with ONE HIP EVENT PAIR PER CALL; the figure is the MEDIAN (the minimum is printed beside it), so a single
    hiccup -- a host garbage collection, a first-use code-object load -- cannot poison it.
    Cache-cold by construction (round 4): every leg rotates over distinct inputs whose total size exceeds the 256 MiB
    Infinity Cache -- 5 arrays for the 64 MiB 1-D configs, 3 for the 256 MiB 2-D ones -- as the headline does; `inputs_rotated`
    is printed with every figure.  C2 / C4 also carry a `roofline` object of their dominant kernel (same schema as the
    headline's)."""
    res = []

    def run(label, L, dtag, xs, mkfn, alg_bytes, roofline=None):
        x = xs[0]
        W.reserve_workspace(x, L, full=True) if _reserve_has_full(W) else W.reserve_workspace(x, L)
        med, mn, mean = _event_each_ms([mkfn(t) for t in xs], reps)
        row = {"workload": label, "L": int(L), "dtype": dtag, "reps": reps, "inputs_rotated": len(xs), "ms_per_step": round(med, 5),
               "ms_min": round(mn, 5), "ms_mean": round(mean, 5), "Msamples_per_s": round(x.numel() / med / 1e3, 1),
               "algorithmic_GBps": round(alg_bytes / med / 1e6, 1), "frac": round(alg_bytes / med / 1e6 / HBM_PEAK_GBPS, 4),
               "kernel": W.last_kernel()}
        if roofline is not None:
            row["roofline"] = roofline(row["kernel"])
        res.append(row)

    for name in ("c1", "c2", "c4"):
        made = [make_workload(W, name, device, 42 + 17 * j) for j in range(NROT_1D)]
        label, _, wt, L, dtag = made[0]
        xs = [m[1] for m in made]
        y = W.similar(xs[0])
        rf = None
        if name in ("c2", "c4"):
            rf = (lambda kern, xs=xs, wt=wt, name=name: roofline_leg(W, xs, wt, False, xs[0].element_size(), 100, kern, tag=name))
        run(label, L, dtag, xs, (lambda t, y=y, wt=wt, L=L: (lambda: W.dwt_oop_(y, t, wt, L))), 2 * xs[0].numel() * xs[0].element_size(), rf)
        del xs, y, made
        torch.cuda.empty_cache()
    # the inverse of the headline config and the section 8(f) rows (3-D, modwt, denoise), same protocol
    g = torch.Generator(device="cpu").manual_seed(7)
    db4 = W.wavelet(W.WT.db4)
    x2s = [torch.randn(8192, 8192, generator=g, dtype=torch.float32).to(device).t() for _ in range(NROT)]
    y2 = W.similar(x2s[0])
    a2 = 2 * x2s[0].numel() * 4
    run("2-D idwt db4 filter 8192x8192 f32", 13, "f32", x2s, lambda t: (lambda: W.idwt_oop_(y2, t, db4, 13)), a2)
    sym5 = W.wavelet(W.WT.sym5)
    run("2-D idwt sym5 (10 taps, the reference's DEFAULT_WAVELET) filter 8192x8192 f32", 13, "f32", x2s,
        lambda t: (lambda: W.idwt_oop_(y2, t, sym5, 13)), a2)
    cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    run("2-D dwt cdf9/7 lifting 8192x8192 f32", 13, "f32", x2s, lambda t: (lambda: W.dwt_oop_(y2, t, cdf, 13)), a2)
    run("2-D idwt cdf9/7 lifting 8192x8192 f32", 13, "f32", x2s, lambda t: (lambda: W.idwt_oop_(y2, t, cdf, 13)), a2)
    sym8 = W.wavelet(W.WT.sym8)
    run("2-D dwt sym8 (16 taps) filter 8192x8192 f32", 13, "f32", x2s, lambda t: (lambda: W.dwt_oop_(y2, t, sym8, 13)), a2)
    run("2-D idwt sym8 (16 taps) filter 8192x8192 f32", 13, "f32", x2s, lambda t: (lambda: W.idwt_oop_(y2, t, sym8, 13)), a2)
    batt6 = W.wavelet(W.WT.batt6)
    run("2-D dwt batt6 (59 taps) filter 8192x8192 f32", 13, "f32", x2s, lambda t: (lambda: W.dwt_oop_(y2, t, batt6, 13)), a2)
    del x2s, y2
    torch.cuda.empty_cache()
    x2d = [torch.randn(8192, 8192, generator=g, dtype=torch.float64).to(device).t() for _ in range(NROT)]
    y2d = W.similar(x2d[0])
    run("2-D dwt db4 filter 8192x8192 f64", 13, "f64", x2d, lambda t: (lambda: W.dwt_oop_(y2d, t, db4, 13)), 2 * x2d[0].numel() * 8)
    run("2-D idwt db4 filter 8192x8192 f64", 13, "f64", x2d, lambda t: (lambda: W.idwt_oop_(y2d, t, db4, 13)), 2 * x2d[0].numel() * 8)
    del x2d, y2d
    torch.cuda.empty_cache()
    x3 = [torch.randn(512, 512, 512, generator=g, dtype=torch.float32).to(device).permute(2, 1, 0) for _ in range(2)]   # 2 x 512 MiB
    y3 = W.similar(x3[0])
    run("3-D dwt db4 filter 512^3 f32", 9, "f32", x3, lambda t: (lambda: W.dwt_oop_(y3, t, db4, 9)), 2 * x3[0].numel() * 4)
    run("3-D idwt db4 filter 512^3 f32", 9, "f32", x3, lambda t: (lambda: W.idwt_oop_(y3, t, db4, 9)), 2 * x3[0].numel() * 4)
    del x3, y3
    torch.cuda.empty_cache()
    x3d = [torch.randn(512, 512, 512, generator=g, dtype=torch.float64).to(device).permute(2, 1, 0) for _ in range(2)]   # 2 x 1 GiB
    y3d = W.similar(x3d[0])
    run("3-D dwt db4 filter 512^3 f64", 9, "f64", x3d, lambda t: (lambda: W.dwt_oop_(y3d, t, db4, 9)), 2 * x3d[0].numel() * 8)
    del x3d, y3d
    torch.cuda.empty_cache()
    xm = [torch.randn(1 << 24, generator=g, dtype=torch.float32).to(device) for _ in range(2)]     # (the 9 x 64 MiB output alone exceeds the cache)
    run("1-D modwt db4 2^24 f32 (output 2^24 x 9)", 8, "f32", xm, lambda t: (lambda: W.modwt(t, db4, 8)), (1 + 9) * xm[0].numel() * 4)
    del xm
    # translation-invariant denoise (denoising.jl:36-67), default wavelet sym5, 8 x 8 spins as one device-resident batch:
    # 64 forward + 64 inverse transforms of the image per call; algorithmic bytes = (read + write) per spin and direction
    xd = [torch.randn(2048, 2048, generator=g, dtype=torch.float32).to(device).t()]
    run("2-D denoise TI 8x8 spins sym5 2048x2048 f32 (64 dwt + 64 idwt, fused batch)", 6, "f32", xd,
        lambda t: (lambda: W.denoise(t, TI=True)), 64 * 2 * 2 * xd[0].numel() * 4)
    del xd
    W.destroy_contexts()
    torch.cuda.empty_cache()
    annotate_secondary(res)
    return res


# un-fused Float32 VALU peak: 256 CUs x 4 SIMDs x 32 lanes per cycle x 2.4 GHz, ONE multiply or add per lane-cycle (the exact mode
# never contracts them); Float64 runs at half that rate.  (The 157.3 TFLOP/s of MI355X_MICROARCH.md counts an FMA as two.)
VALU_PEAK_F32_TFLOPS = 256 * 4 * 32 * 2.4e9 / 1e12
SECONDARY_PMC = os.path.join(ROOT, "profiles", "r06_secondary_pmc.json")


def annotate_secondary(rows):
    """Every secondary row gets a `roofline` object {bound, frac, peak, unit, traffic, counters} (round-5 review item 3).  `bound` is
    "valu" where the arithmetic intensity says so (>= 16 taps, the 59-tap batt6, modwt's Float64-rate taps): frac = algorithmic flops
    (2F - 1 per output and pass, un-fused) / time / the un-fused VALU peak; "hbm" otherwise: frac = algorithmic bytes / time / 8 TB/s.
    `traffic` and `counters` (VALU busy, HBM rate, LDS conflicts of the dominant kernel) come from the committed rocprofv3 PMC
    collection profiles/r06_secondary_pmc.json (tools/r06_secondary_pmc.sh); `counters_read_as` says what they point at when that is
    not the stated bound."""
    pmc = {}
    try:
        for r in json.load(open(SECONDARY_PMC)):
            pmc[r["case"]] = r
    except (OSError, ValueError):
        pass
    taps = {"sym8": 16, "batt6": 59, "sym5": 10, "db4": 8}
    case_of = [("C2", "c2"), ("1-D dwt db4 filter 2^24", "c2"), ("1-D dwt cdf9/7", "c4"), ("3-D dwt db4 filter 512^3 f32", "dwt3d"), ("3-D idwt db4 filter 512^3 f32", "idwt3d"), ("2-D dwt cdf9/7", "lift2d"),
               ("2-D idwt cdf9/7", "lift2d_inv"), ("2-D dwt sym8", "sym8_fwd"), ("2-D idwt sym8", "sym8_inv"), ("1-D modwt", "modwt"),
               ("2-D dwt batt6", "batt6")]
    for row in rows:
        if isinstance(row.get("roofline"), dict) and "bound" in row["roofline"]:
            continue
        lab = row["workload"]
        case = next((c for pre, c in case_of if lab.startswith(pre)), None)
        p = pmc.get(case, {})
        n = {"8192x8192": 8192 * 8192, "512^3": 512 ** 3, "2^24": 1 << 24, "2048x2048": 2048 * 2048}
        nsamp = next((v for k, v in n.items() if k in lab), None)
        F = next((v for k, v in taps.items() if k in lab), None)
        rf = {"bound": "hbm", "frac": row["frac"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "achieved": row["algorithmic_GBps"]}
        valu_case = (F is not None and F >= 16 and "filter" in lab) or lab.startswith("1-D modwt")
        if valu_case and nsamp:
            if lab.startswith("1-D modwt"):
                flops = nsamp * row["L"] * 2 * (2 * 8 - 1)               # every level: s and d for every sample, 8 taps, Float64 arithmetic
                peak = VALU_PEAK_F32_TFLOPS / 2
            else:
                flops = nsamp * 2 * (2 * F - 1) * (4.0 / 3.0)              # 2 passes per level, levels shrink by 4
                peak = VALU_PEAK_F32_TFLOPS
            ach = flops / (row["ms_per_step"] * 1e-3) / 1e12
            rf = {"bound": "valu", "frac": round(ach / peak, 4), "peak": round(peak, 1), "unit": "TFLOP/s (un-fused mul + add)", "achieved": round(ach, 2)}
        if p:
            rf["traffic"] = p.get("traffic_bytes")
            rf["traffic_kernel"] = p.get("kernel")
            rf["counters"] = {k: p[k] for k in ("valu_busy", "hbm_TBps_under_pmc", "issue_stall", "lds_conflict", "traffic_over_algorithmic") if k in p}
            if p.get("bound") and p["bound"] != rf["bound"]:
                rf["counters_read_as"] = p["bound"]
        else:
            rf["traffic"] = None
        row["roofline"] = rf


def reference_shapes_leg(W, device, reps=20):
    """The reference's own GPU benchmark (benchmark/gpu_benchmark.jl:57-67,70-80,83-96,98-106,109-118,138-148; BASELINE.md
    section 1): 1-D db4 dwt / idwt of 2^16, 2^18, 2^20 f32 with L = min(8, max); 1-D wpt / iwpt of 2^14, 2^16, 2^18 (full
    tree); 2-D 512 / 1024 / 2048 with L = min(4, max); 3-D 32 / 64 / 128 with L = min(3, max); 1-D cdf9/7 lifting 2^16 ... 2^20.
    Like the reference's harness these go through the ALLOCATING calls (dwt(x, wt, L): `similar` + transform) on ONE resident
    input -- sizes at which a transform is launch-latency, not bandwidth: the figure that matters is microseconds per call."""
    g = torch.Generator(device="cpu").manual_seed(11)
    db4 = W.wavelet(W.WT.db4)
    cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    rows = []

    def run(label, x, fn, L):
        med, mn, _ = _event_each_ms([fn], reps, warm=5, warm_ms=10.0)
        rows.append({"case": label, "L": L, "us_per_call": round(med * 1e3, 2), "us_min": round(mn * 1e3, 2),
                     "Msamples_per_s": round(x.numel() / med / 1e3, 1), "kernel": W.last_kernel()})

    for pw in (16, 18, 20):
        x = torch.randn(1 << pw, generator=g, dtype=torch.float32).to(device)
        L = min(8, W.maxtransformlevels(x))
        run(f"1D filter dwt 2^{pw}", x, lambda: W.dwt(x, db4, L), L)
        yc = W.dwt(x, db4, L)
        run(f"1D filter idwt 2^{pw}", x, lambda: W.idwt(yc, db4, L), L)
        run(f"1D lifting dwt 2^{pw}", x, lambda: W.dwt(x, cdf, L), L)
        yl = W.dwt(x, cdf, L)
        run(f"1D lifting idwt 2^{pw}", x, lambda: W.idwt(yl, cdf, L), L)
    for pw in (14, 16, 18):
        x = torch.randn(1 << pw, generator=g, dtype=torch.float32).to(device)
        run(f"1D filter wpt 2^{pw}", x, lambda: W.wpt(x, db4), W.maxtransformlevels(x))
        wp = W.wpt(x, db4)
        run(f"1D filter iwpt 2^{pw}", x, lambda: W.iwpt(wp, db4), W.maxtransformlevels(x))
    for n in (512, 1024, 2048):
        x = torch.randn(n, n, generator=g, dtype=torch.float32).to(device).t()
        L = min(4, W.maxtransformlevels(x))
        run(f"2D filter dwt {n}x{n}", x, lambda: W.dwt(x, db4, L), L)
        yc = W.dwt(x, db4, L)
        run(f"2D filter idwt {n}x{n}", x, lambda: W.idwt(yc, db4, L), L)
    for n in (32, 64, 128):
        x = torch.randn(n, n, n, generator=g, dtype=torch.float32).to(device).permute(2, 1, 0)
        L = min(3, W.maxtransformlevels(x))
        run(f"3D filter dwt {n}^3", x, lambda: W.dwt(x, db4, L), L)
        yc = W.dwt(x, db4, L)
        run(f"3D filter idwt {n}^3", x, lambda: W.idwt(yc, db4, L), L)
    return {"source": "benchmark/gpu_benchmark.jl shapes (randn Float32, db4 / cdf9/7), allocating calls, one resident input, median of "
                      f"{reps} event-timed calls", "rows": rows}


def _reserve_has_full(W):
    import inspect
    try:
        return "full" in inspect.signature(W.reserve_workspace).parameters
    except (TypeError, ValueError):
        return False


def _rocprof_summary(kname_substr, tag="c3"):
    """Average duration of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary of this same
    command (profiles/<round>_<tag>_kernel_stats.csv), so that the line can be checked against profiles/ without a GPU."""
    import csv
    rel = os.path.join("profiles", f"{PROFILE_ROUND}_{tag}_kernel_stats.csv")
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None
    best = None
    for r in csv.DictReader(open(path)):
        if kname_substr in r.get("Name", ""):
            if best is None or float(r["AverageNs"]) > float(best["AverageNs"]):
                best = r
    if best is None:
        return None
    return {"file": rel, "kernel": best["Name"], "calls": int(best["Calls"]),
            "avg_launch_ms": round(float(best["AverageNs"]) * 1e-6, 5)}


# levels finished by ONE launch of the kernel that consumes the full-size input (an L = Ldom call is exactly that launch)
DOMINANT_LEVELS = {"k_fwd2d_stream2": 2, "k_fwd2d_pair": 2, "k_fwd2d_pair64": 2, "k_fwd1d_multi": 4, "k_lift1d_fwd3": 3}


def _measure_traffic_live(tag):
    """HBM bytes per launch of the dominant kernel of config `tag`, measured NOW with rocprofv3 (--pmc FETCH_SIZE and --pmc
    WRITE_SIZE in separate passes with --kernel-trace only, through the torch-free harness tools/wlbench.bin), FETCH_SIZE
    doubled (MI355X_MICROARCH.md: gfx950 counts half of the 16-B/lane coalesced reads).  None when rocprofv3 / the harness
    are not there or a pass fails -- the caller then falls back to the committed collection and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("WL_BENCH_NO_LIVE_PMC") == "1" or shutil.which("rocprofv3") is None:
        return None
    harness = os.path.join(ROOT, "tools", "wlbench.bin")
    rp = os.path.join(ROOT, "tools", "rp.sh")
    if not (os.path.exists(harness) and os.path.exists(rp)):
        return None
    cases = {"c3": (["L=2"], "k_fwd2d_pair<8, 2, 1, 0"),
             "c2": (["n0=16777216", "n1=1", "L=4"], "k_fwd1d_multi<float, 8, 1>"),
             "c5": (["dwtc=1", "n0=65536", "n1=8192", "L=4"], "k_fwd1d_multi<float, 8, 1>")}
    if tag not in cases:
        return None
    argv, head = cases[tag]
    out = {}
    tmp = tempfile.mkdtemp(prefix="wl_pmc_")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "wavelets.jl_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    env["RP_MAX_ITERS"] = "80"
    try:
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, name)
            subprocess.run(["bash", rp, d, "live", f"--kernel-trace --pmc {name}", harness] + argv + ["reps=12", "warm=3", "check=0"],
                           env=env, timeout=90, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for h in hits:
                for r in csv.DictReader(open(h)):
                    if head in r["Kernel_Name"] and r["Counter_Name"] == name:
                        vals.append(float(r["Counter_Value"]))
            if len(vals) < 4:
                return None
            out[name] = (sum(vals) / len(vals), len(vals))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch = out["FETCH_SIZE"][0] * 1024 * 2
    write = out["WRITE_SIZE"][0] * 1024
    return {"hbm_bytes_per_launch": int(fetch + write), "fetch_bytes_corrected": int(fetch), "write_bytes": int(write),
            "launches": [out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]], "kernel": head}


def roofline_leg(W, xs, wt, batched, esize, reps, main_kernel, tag="c3", live_pmc=False):
    """Dominant kernel = the launch that consumes the full-size input.  Its algorithmic bytes are
    2*N*sizeof(T): it reads every input sample once and writes N coefficients (8 B/sample
    f32) -- that holds for the single-level kernels and for the fused kernels, which finish several levels in the same pass
    (pair: 3/4 N level-1 details + 1/4 N level-2 coefficients; the level-1 approximation never leaves the chip).
    A call with L = 1 (L = 2 for the fused pair; L = 3 for the fused lifting line kernel; L = 4 for the multi-level line kernel) is exactly
    one launch of that kernel (its first-level template instance, which rocprofv3 --stats reports under its own name).
    `frac` uses HIP events on the launch stream around a train of such launches rotating over the inputs (live, this run);
    `rocprof` repeats the computation from the committed rocprofv3 summary; `traffic` = HBM bytes per launch from the PMC
    counters: measured in this run when rocprofv3 is present (live_pmc), otherwise the committed collection."""
    Ldom = DOMINANT_LEVELS.get(main_kernel, 1)
    x = xs[0]
    y1 = W.similar(x)
    mk = (lambda t: (lambda: W.dwtc_(y1, t, wt, Ldom))) if batched else (lambda t: (lambda: W.dwt_oop_(y1, t, wt, Ldom)))
    fns = [mk(t) for t in xs]
    train_ms = _event_train_ms(fns, reps)
    med_ms, min_ms, avg_ms = _event_each_ms(fns, reps, warm=3)
    kname = W.last_kernel()
    alg_bytes = 2 * x.numel() * esize
    achieved = alg_bytes / (train_ms * 1e-3) / 1e9
    traffic, note = None, "traffic: no PMC summary committed for this kernel yet"
    live = _measure_traffic_live(tag) if live_pmc else None
    if live is not None:
        traffic = live["hbm_bytes_per_launch"]
        note = ("measured in THIS run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (--kernel-trace only) of "
                f"tools/wlbench.bin on the same kernel ({live['kernel']}, {live['launches'][0]} / {live['launches'][1]} launches); FETCH_SIZE x2 per "
                f"MI355X_MICROARCH.md (gfx950); fetch {live['fetch_bytes_corrected']} B + write {live['write_bytes']} B; "
                f"traffic/algorithmic = {traffic / alg_bytes:.3f}")
    else:
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json" if tag == "c3" else f"pmc_{tag}.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("kernel_short") == kname:
                    traffic = j.get("hbm_bytes_per_launch")
                    note = j.get("note", "") + f" (static: read from the committed profiles/{os.path.basename(pmc)}, not measured by this run)"
            except Exception:
                pass
    levels = {1: "level 1", 2: "levels 1-2", 3: "levels 1-3", 4: "levels 1-4"}[Ldom]
    out = {"bound": "hbm", "kernel": f"{kname} (first launch: {levels})",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
           "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(train_ms, 5),
           "timing": f"HIP events on the launch stream inside a back-to-back train of {reps} launches (median of chunks of 10), {len(xs)} inputs in rotation",
           "isolated_launch_ms": {"avg": round(avg_ms, 5), "median": round(med_ms, 5), "min": round(min_ms, 5),
                                  "note": "one event pair per launch"},
           "launches_timed": reps, "traffic_note": note,
           "frac_of_measured_copy_6290GBps": round(achieved / 6290.0, 4)}
    rp = _rocprof_summary(kname, tag)
    if rp is not None:
        rp["achieved"] = round(alg_bytes / (rp["avg_launch_ms"] * 1e-3) / 1e9, 1)
        rp["frac"] = round(rp["achieved"] / HBM_PEAK_GBPS, 4)
        out["rocprof"] = rp
    del y1
    return out


def cpu_baseline_leg(W, workload, wt, L):
    """The reference's CPU path, represented by the oracle's sources (literal C restatement of the reference's
    single-threaded loops) compiled -O3 -march=native -ffp-contract=off ON THIS HOST and timed around
    the C call itself: 1 warm-up + 3 repetitions, median.  `value` is the 1-core figure (the reference has no threading); the
    OpenMP-over-lines variant on all host cores is reported beside it for the 2-D case."""
    import oracle                      # the checker / baseline -- never the product path
    h = oracle.build_native()
    rng = np.random.default_rng(42)
    ncores = os.cpu_count() or 1
    extra = {}
    med = statistics.median
    if workload == "c3":
        n = 8192
        xf = np.asfortranarray(rng.standard_normal((n, n), dtype=np.float32))
        # warm-up on a quarter-size array (page faults, clocks), then the full 8192 x 8192 transform three times
        oracle.time_dwt_filter(h, np.asfortranarray(xf[:4096, :4096]), wt.qmf, 12, reps=1, warmup=0)
        ts = oracle.time_dwt_filter(h, xf, wt.qmf, L, reps=3, warmup=0)
        dt, ns = med(ts), xf.size
        sample = (f"the full 2-D db4 dwt of the {n}x{n} f32 array, L={L}: 1 warm-up (4096x4096) + 3 repetitions, median "
                  f"(min {min(ts):.2f} s, max {max(ts):.2f} s), 1 thread")
        nt = max(1, min(ncores, int(h.wlo_max_threads())))
        tm = oracle.time_dwt_filter(h, xf, wt.qmf, L, reps=3, warmup=1, threads=nt)
        extra["all_cores"] = {"value": round(ns / med(tm) / 1e6, 2), "unit": "Msamples/s", "cores": nt, "seconds": round(med(tm), 3),
                              "note": "same loops, the independent lines of every level spread over OpenMP threads "
                                      "(bit-identical result; the reference itself is single-threaded)"}
        # anchor against the one number the reference publishes for this path (README.md:249-250: 1-D db2 dwt of 2^20
        # Float64, 20 levels, 24.8 ms per call on unstated hardware): the same call through the same build on this host
        x1 = np.asfortranarray(rng.random(1 << 20))
        db2 = W.wavelet(W.WT.db2)
        t1 = oracle.time_dwt_filter(h, x1, db2.qmf, 20, reps=5, warmup=1)
        extra["c1_anchor"] = {"workload": "1-D dwt db2 filter 2^20 f64, L=20, 1 thread, 1 warm-up + 5 repetitions, median",
                              "oracle_ms_per_call": round(med(t1) * 1e3, 2), "reference_readme_ms_per_call": 24.8,
                              "reference_hardware": "unstated (README.md:249-250)"}
    else:
        if workload == "c1":
            xf, what = np.asfortranarray(rng.standard_normal(1 << 20)), "full-size 1-D db2 f64 transform"
        elif workload == "c2":
            xf, what = np.asfortranarray(rng.standard_normal(1 << 24, dtype=np.float32)), "full-size 1-D db4 f32 transform"
        elif workload == "c4":
            xf, what = None, "full-size 1-D cdf9/7 lifting transform"
        else:
            xf, what = None, "1/64 sub-batch: 1024 of the 65536 signals (column-wise 1-D db4 transforms of length 2^16), scaled linearly"
        if workload == "c4":
            xs = rng.standard_normal(1 << 24, dtype=np.float32)
            oracle.dwt_lifting(xs, wt, L)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                oracle.dwt_lifting(xs, wt, L)
                ts.append(time.perf_counter() - t0)
            dt, ns = med(ts), xs.size
            what += " (through the Python wrapper of the -O2 test build: includes two array copies)"
        elif workload == "c5":
            # columns are independent 1-D transforms: time them as a (2^16 x 1024) batch of lines, one after the other
            xcols = np.asfortranarray(rng.standard_normal((1 << 16, 1024), dtype=np.float32))
            def one_pass():
                t0 = time.perf_counter()
                for j in range(0, xcols.shape[1], 1):
                    oracle.time_dwt_filter(h, xcols[:, j], wt.qmf, L, reps=1, warmup=0)
                return time.perf_counter() - t0
            one_pass()
            ts = [one_pass() for _ in range(3)]
            dt, ns = med(ts), xcols.size
        else:
            ts = oracle.time_dwt_filter(h, xf, wt.qmf, L, reps=3, warmup=1)
            dt, ns = med(ts), xf.size
        sample = f"{what}, 1 warm-up + 3 repetitions, median, 1 thread"
    out = {"value": round(ns / dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "build": "gcc -O3 -march=native -ffp-contract=off (oracle sources, built on this host)",
           "sample": sample, "seconds": round(dt, 3), "host_cores_available": ncores}
    out.update(extra)
    return out


if __name__ == "__main__":
    main()
