"""RCCL on the driver's path: the two multi-GPU hosts run their collectives for real on however many HIP devices the box has
(one on the round-end 1-GPU lease -- ncclCommInitAll / init_process_group, the broadcast of the wavelet and the all-reduces
still execute; on an 8-GPU node the same tests are the N = 8 check).

  * tools/wlbench_mgpu.bin (native: one process, one thread + context per device, ncclCommInitAll, ncclBroadcast,
    ncclAllReduce): its all-reduced checksum against the oracle's on the same synthetic shards;
  * bench.py --workload c5 (one process per GPU, torch.distributed backend "nccl" = RCCL): the rank count its all-reduce
    returns equals the device count, every signal is covered once.
The path has no data-path collective: what crosses GPUs is the 2 KiB wavelet description and 8-byte reductions.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "wlbench_mgpu.bin")


def _fill(n, seed):
    """tools/wlbench_mgpu.cpp k_fill on the host: splitmix64 of (i + 1) * golden + seed -> [-0.5, 0.5) in Float32"""
    i = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).astype(np.float32) - np.float32(0.5)


def test_native_host_runs_rccl_on_every_device(gpu, W, oracle):
    import torch
    from wavelets_jl_amd import sharding
    ndev = torch.cuda.device_count()
    length, L = 4096, 12
    signals = 1024 * ndev + 3                                   # ragged: the shards are not all equal
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([BIN, f"gpus={ndev}", f"signals={signals}", f"len={length}", f"L={L}", "steps=2", "warmup=1"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == ndev and d["rccl"]["ranks"] == ndev
    assert d["config"]["signals_covered_all_ranks"] == signals
    assert d["config"]["taps_received_by_last_rank"] == 8       # the last rank built its taps from the broadcast
    wt = W.wavelet(W.WT.db4)
    q = np.array([0.23037781330889648, 0.7148465705529157, 0.6308807679298589, -0.027983769416860003, -0.18703481171909309,
                  0.030841381835560722, 0.03288301166688518, -0.010597401785069035])     # the host's own db4 table (wlbench_mgpu.cpp)
    assert np.abs(q - wt.qmf).max() < 1e-15
    total = 0.0
    for r in range(ndev):
        lo, hi = sharding.shard_range(signals, r, ndev)
        x = _fill((hi - lo) * length, 4242 + 1000 * r).reshape(hi - lo, length).T          # len x ncol, column-major
        y = oracle.dwtc_filter(np.asfortranarray(x), q, L)
        total += float(y.astype(np.float64).sum())
    n = signals * length
    assert abs(d["checksum_all_ranks"] - total) <= 1e-9 * np.sqrt(n) + 1e-7 * abs(total), (d["checksum_all_ranks"], total)


def test_bench_c5_runs_over_rccl_on_every_device(gpu):
    import torch
    ndev = torch.cuda.device_count()
    signals = 2048 * ndev
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", WL_BENCH_FORCE_DIST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev), "--workload", "c5", "--steps", "2", "--warmup", "1",
                          "--c5-signals", str(signals), "--no-cpu"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ndev
    assert d["rccl"]["backend"].startswith("nccl") and d["rccl"]["ranks_counted_by_allreduce"] == ndev, d["rccl"]
    assert d["c5_signals_covered"] == signals and d["config"]["signals_total"] == signals
    assert d["config"]["kernel"].startswith("k_fwd1d")
