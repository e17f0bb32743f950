"""N > 1 host logic on CPU: two processes, gloo backend.  The per-rank transform is injected, so
the test exercises partitioning, the wavelet broadcast from rank 0 and the end-of-run reductions
with the oracle standing in for the GPU kernels (tests may use the oracle; the product never does)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import wavelets_jl_amd as W
    from wavelets_jl_amd import sharding
    cpu = torch.device("cpu")
    nsig, n, L = 10, 256, 5
    rng = np.random.default_rng(123)
    X = rng.standard_normal((n, nsig)).astype(np.float32)         # every rank builds the same batch
    lo, hi = sharding.shard_range(nsig, rank, world)
    # only rank 0 knows the real wavelet; the others start with a decoy and must receive rank 0's
    wt = W.wavelet(W.WT.db4) if rank == 0 else W.wavelet(W.WT.haar)
    y, checksum = sharding.sharded_columnwise(lambda xl, w, l: oracle.dwtc_filter(xl, w.qmf, l),
                                              np.ascontiguousarray(X[:, lo:hi]), wt, L, dist, cpu)
    # lifting scheme broadcast as well
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting) if rank == 0 else W.wavelet(W.WT.haar, W.WT.Lifting)
    sch = sharding.broadcast_wavelet(sch, dist, cpu)
    tmax = sharding.max_over_ranks(1.0 + rank, dist, cpu)
    q.put((rank, lo, hi, y, checksum, len(sch.step), sch.norm1, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_batch():
    sys.path.insert(0, ROOT)
    import oracle
    import wavelets_jl_amd as W
    oracle.build()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(123)
    X = rng.standard_normal((256, 10)).astype(np.float32)
    full = oracle.dwtc_filter(X, W.wavelet(W.WT.db4).qmf, 5)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 5, 5, 10)
    got = np.concatenate([res[0][3], res[1][3]], axis=1)
    assert np.array_equal(got, full)                               # rank 1 used rank 0's taps, bit for bit
    assert abs(res[0][4] - float(full.astype(np.float64).sum())) < 1e-6 * abs(full).sum()
    assert res[0][4] == res[1][4]
    assert res[1][5] == 4 and res[1][6] == 1.1496043988603355      # cdf9/7 scheme arrived intact
    assert res[0][7] == 2.0 and res[1][7] == 2.0


def test_shard_range_partition():
    sys.path.insert(0, ROOT)
    from wavelets_jl_amd import sharding
    for n in (1, 7, 8, 65536, 100):
        for world in (1, 2, 3, 4, 8):
            parts = [sharding.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(65536, 3, 8) == (3 * 8192, 4 * 8192)


def test_wavelet_pack_roundtrip():
    sys.path.insert(0, ROOT)
    import wavelets_jl_amd as W
    from wavelets_jl_amd import sharding
    for wt in (W.wavelet(W.WT.db4), W.wavelet(W.WT.batt6)):
        r = sharding.unpack_wavelet(sharding.pack_wavelet(wt))
        assert np.array_equal(r.qmf, wt.qmf)
    for nm in ("cdf97", "db2", "haar"):
        s = W.wavelet(getattr(W.WT, nm), W.WT.Lifting)
        r = sharding.unpack_wavelet(sharding.pack_wavelet(s))
        a, b = s.flatten(), r.flatten()
        assert all(np.array_equal(u, v) for u, v in zip(a, b)) and (r.norm1, r.norm2) == (s.norm1, s.norm2)
