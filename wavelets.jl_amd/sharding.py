"""Multi-GPU host logic for the batched (independent signals / images) workloads.

The transform path has no exchange step: independent units (columns of a len x nsignals matrix,
or whole images) are sharded across ranks, one process per GPU, and every rank runs the same
single-GPU kernels on its shard.  The only collectives are (1) a broadcast of the wavelet
description (filter taps or lifting steps: tens of bytes) from rank 0 -- RCCL over xGMI when the
backend is "nccl" -- and (2) a MAX all-reduce of the elapsed time / a checksum at the end.
A single signal is never split across GPUs (north_star).
"""
from __future__ import annotations

import numpy as np
import torch

from .wt import GLS, LSStep, LSStepParam, OrthoFilter, Predict, Update, UpdateStep

_PACK_LEN = 256


def shard_range(nunits: int, rank: int, world: int):
    """Contiguous block partition of `nunits` independent units: rank r owns [lo, hi)."""
    base, rem = divmod(int(nunits), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_wavelet(wt) -> np.ndarray:
    v = np.zeros(_PACK_LEN, dtype=np.float64)
    if isinstance(wt, OrthoFilter):
        v[0], v[1] = 0.0, len(wt.qmf)
        v[2:2 + len(wt.qmf)] = wt.qmf
    elif isinstance(wt, GLS):
        v[0], v[1], v[2], v[3] = 1.0, len(wt.step), wt.norm1, wt.norm2
        p = 4
        for s in wt.step:
            v[p] = 1.0 if isinstance(s.steptype, UpdateStep) else 0.0
            v[p + 1] = len(s.param.coef)
            v[p + 2] = s.param.shift
            v[p + 3:p + 3 + len(s.param.coef)] = s.param.coef
            p += 3 + len(s.param.coef)
    else:
        raise TypeError("wavelet must be an OrthoFilter or a GLS")
    return v


def unpack_wavelet(v: np.ndarray, name: str = "broadcast"):
    v = np.asarray(v, dtype=np.float64)
    if int(v[0]) == 0:
        n = int(v[1])
        return OrthoFilter(v[2:2 + n].copy(), name_=name)
    nsteps = int(v[1])
    steps, p = [], 4
    for _ in range(nsteps):
        nc = int(v[p + 1])
        steps.append(LSStep(LSStepParam(v[p + 3:p + 3 + nc].copy(), int(v[p + 2])), Update if v[p] else Predict))
        p += 3 + nc
    return GLS((steps, float(v[2]), float(v[3]), name))


def broadcast_wavelet(wt, dist, device):
    """Rank 0's wavelet description to every rank (no-op without a process group)."""
    if dist is None or not dist.is_initialized():
        return wt
    rank = dist.get_rank()
    buf = torch.from_numpy(pack_wavelet(wt) if rank == 0 else np.zeros(_PACK_LEN)).to(device)
    dist.broadcast(buf, src=0)
    return unpack_wavelet(buf.cpu().numpy(), getattr(wt, "name", "broadcast"))


def max_over_ranks(value: float, dist, device) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist, device) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def sharded_columnwise(transform, x_local, wt, L, dist, device):
    """Run `transform(x_local, wt, L)` (e.g. wavelets_jl_amd.dwtc) on this rank's column shard with
    rank 0's wavelet; returns (y_local, global checksum).  No signal data crosses ranks."""
    wt = broadcast_wavelet(wt, dist, device)
    y = transform(x_local, wt, L)
    local = float(torch.as_tensor(y, dtype=torch.float64).sum()) if not isinstance(y, np.ndarray) else float(y.sum())
    return y, sum_over_ranks(local, dist, device)
