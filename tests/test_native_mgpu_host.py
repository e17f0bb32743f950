"""The native (torch-free) multi-GPU host tools/wlbench_mgpu.cpp: builds, links against librccl + the product library, and its
host-only `dry=1` mode prints the column partition (wl_shard_range, the same arithmetic as sharding.shard_range).  The device
run (1 rank through gpurun, N ranks on an 8-GPU node) is recorded in profiles/; this test needs no GPU."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "wlbench_mgpu.bin")


def test_native_host_builds_links_and_partitions():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "wlbench_mgpu.bin"])
    needed = subprocess.run(["readelf", "-d", BIN], capture_output=True, text=True, check=True).stdout
    assert "librccl.so" in needed and "libwavelets_mi355x.so" in needed
    syms = subprocess.run(["nm", "-D", "--undefined-only", BIN], capture_output=True, text=True, check=True).stdout
    for s in ("ncclCommInitAll", "ncclBroadcast", "ncclAllReduce", "wl_dwtc_filter", "wl_shard_range", "wl_ctx_create"):
        assert s in syms, s
    from wavelets_jl_amd import sharding
    for world, signals in ((8, 65536), (3, 1000), (5, 7)):
        out = subprocess.run([BIN, "dry=1", f"gpus={world}", f"signals={signals}"], capture_output=True, text=True, check=True).stdout
        d = json.loads(out)
        assert d["dry"] and d["n_gpus"] == world and d["signals_covered"] == signals
        assert [tuple(s) for s in d["shards"]] == [sharding.shard_range(signals, r, world) for r in range(world)]
