"""bench.py honours --gpus N: started without a launcher it re-launches itself through torch.distributed.run with N
ranks (here: 2 CPU ranks on the gloo backend with the transform stubbed out -- the launcher, rendezvous, shard partition
and reductions are what is under test; nothing is measured)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_respawn_command_line():
    sys.path.insert(0, ROOT)
    import bench
    import argparse
    saved = os.environ.pop("WORLD_SIZE", None)
    try:
        ns = argparse.Namespace(gpus=4)
        cmd = bench.respawn_command(ns, ["--gpus", "4", "--steps", "3"])
        assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
        assert bench.respawn_command(argparse.Namespace(gpus=1), []) is None
        os.environ["WORLD_SIZE"] = "4"
        assert bench.respawn_command(ns, []) is None            # already under a launcher: never re-launch
    finally:
        os.environ.pop("WORLD_SIZE", None)
        if saved is not None:
            os.environ["WORLD_SIZE"] = saved


def test_gpus_flag_spawns_that_many_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--stub-backend", "gloo"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["gpus_requested"] == 2 and out["stub"] is True
    assert out["c5_signals_covered"] == 65536.0                  # the two shards partition the batch
    assert out["checksum_all_ranks"] == 64 * 8 * (1 + 2)        # rank r's stub shard holds r + 1
    # with more than one GPU the sharded C5 batch is the top-level metric (BASELINE.json configs[4]); the C3 figure is nested
    assert "batched column-wise" in out["metric"] and out["scaling"] == "strong" and out["unit"] == "Msamples/s"
    assert out["config"]["signals_per_rank"] == 32768 and "65536" in out["config"]["workload"]
    assert out["steps"] == 3 and out["ms_per_step"] > 0 and out["higher_is_better"] is True
    assert out["c3_weak_scaling"]["scaling"] == "weak" and out["c3_weak_scaling"]["ms_per_step"] > 0
    assert out["rccl"]["ranks_counted_by_allreduce"] == 2       # counted by an all-reduce of ones, not by torch's bookkeeping
    assert "cpu_baseline" in out


def test_eight_ranks_partition_and_reduce():
    """The N = 8 line of the scaling curve, as far as a host without eight GPUs can run it: eight gloo ranks with the transform
    stubbed out -- 65536 / 8 = 8192 signals per rank, the rank count comes from an all-reduce, the strong-scaling top level and the
    nested weak-scaling C3 object are both there -- and the native host's partition for eight devices (dry run)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--stub-backend", "gloo"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["gpus_requested"] == 8 and out["stub"] is True
    assert out["config"]["signals_per_rank"] == 8192 and out["config"]["signals_total"] == 65536
    assert out["c5_signals_covered"] == 65536.0
    assert out["checksum_all_ranks"] == 64 * 8 * sum(range(1, 9))
    assert out["rccl"]["ranks_counted_by_allreduce"] == 8
    assert out["scaling"] == "strong" and out["c3_weak_scaling"]["scaling"] == "weak" and out["c3_weak_scaling"]["ms_per_step"] > 0
    binp = os.path.join(ROOT, "tools", "wlbench_mgpu.bin")
    if os.path.exists(binp):
        d = json.loads(subprocess.run([binp, "dry=1", "gpus=8"], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        assert d["dry"] and d["n_gpus"] == 8 and d["signals_covered"] == 65536
        assert d["shards"] == [[8192 * r, 8192 * (r + 1)] for r in range(8)]


def test_rank_pinning_reads_the_gpu_local_cpulist(tmp_path):
    """bench.py pins each rank process of an N > 1 run to the cores sysfs lists next to its GPU (local_cpulist); without the file,
    or when the list does not narrow the allowed set, it leaves the affinity alone and says why."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench.parse_cpulist("5") == [5] and bench.parse_cpulist("") == []
    f = bench.local_cpulist_file(0, 0x1b, 0, str(tmp_path))
    assert f.endswith("bus/pci/devices/0000:1b:00.0/local_cpulist")
    miss = bench.pin_rank_to_gpu_cores((0, 0x1b, 0), sysfs=str(tmp_path), apply=False)
    assert miss["pinned"] is False and "reason" in miss
    os.makedirs(os.path.dirname(f))
    allowed = sorted(os.sched_getaffinity(0))
    with open(f, "w") as fh:
        fh.write(f"{allowed[0]}\n")
    hit = bench.pin_rank_to_gpu_cores((0, 0x1b, 0), sysfs=str(tmp_path), apply=False)
    if len(allowed) > 1:
        assert hit == {"pinned": True, "source": f, "cpus": 1, "first": allowed[0], "last": allowed[0]}
    with open(f, "w") as fh:
        fh.write(",".join(str(c) for c in allowed) + "\n")
    same = bench.pin_rank_to_gpu_cores((0, 0x1b, 0), sysfs=str(tmp_path), apply=False)
    assert same["pinned"] is False
