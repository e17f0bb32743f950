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
