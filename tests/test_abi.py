"""The C ABI: the shared library loads without a GPU and exports exactly the symbols
include/wavelets_mi355x.h declares.  No compute call is made here."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wavelets_mi355x.h")


def declared_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"WL_API[^;(]*?\b(wl_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(W):
    lib = W._lib.load()
    syms = declared_symbols()
    assert len(syms) >= 18, syms
    out = subprocess.check_output(["nm", "-D", "--defined-only", W._lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (wl_[a-z0-9_]+)", out))
    assert set(syms) == exported, (set(syms) ^ exported)
    assert set(W._lib.SIGNATURES) == set(syms)
    for s in syms:
        assert getattr(lib, s) is not None


def test_fused_variant_exports_the_same_abi(W):
    """libwavelets_mi355x_fma.so (the opt-in fused arithmetic mode, `make FMA=1`) is the same ABI built from the same sources"""
    path = W._lib.LIB_PATHS["fused"]
    assert os.path.exists(path), "build() makes both libraries"
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    assert set(re.findall(r" T (wl_[a-z0-9_]+)", out)) == set(declared_symbols())
    assert W.get_arithmetic() == "exact"                       # the default; switching needs no device
    try:
        W.set_arithmetic("fused")
        lib = W._lib.load()
        assert lib._name == path and lib.wl_version() == C.CDLL(W._lib.LIB_PATH).wl_version()
    finally:
        W.set_arithmetic("exact")
    assert W._lib.load()._name == W._lib.LIB_PATH


def test_hostonly_entry_points(W):
    lib = W._lib.load()
    assert lib.wl_version() == 100
    assert lib.wl_strerror(-1) == b"size must have a sufficient power of 2 factor"
    assert lib.wl_strerror(-3) == b"in array is out array"
    for n, e in ((1, 0), (2, 1), (40, 3), (1 << 24, 24), (8192, 13), (7, 0)):
        assert lib.wl_maxtransformlevels(n) == e
    dims = (C.c_int64 * 3)(8192, 8192, 1)
    # the fast filter-bank path of C3 holds two approximation buffers of N/4 elements: 128 MiB (+ padding), not 4 N
    wsb = lib.wl_workspace_bytes(0, 2, dims, 13)
    assert 2 * (8192 * 8192 // 4) * 4 <= wsb <= 129 * 2 ** 20
    # ... and the upper bound over every entry point (lifting, long filters, generic kernels, wpt) is about 4.5 N elements
    full = lib.wl_workspace_bytes_full(0, 2, dims, 13)
    assert 4 * 8192 * 8192 * 4 <= full <= 5 * 8192 * 8192 * 4
    # block partition of a batch over ranks (the C helper a Julia host uses; sharding.shard_range is the same arithmetic)
    from wavelets_jl_amd import sharding
    for nunits, world in ((65536, 8), (65536, 3), (10, 4), (3, 8), (0, 2)):
        covered = 0
        for r in range(world):
            lo, hi = C.c_int64(), C.c_int64()
            assert lib.wl_shard_range(nunits, r, world, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == sharding.shard_range(nunits, r, world) and lo.value == covered
            covered = hi.value
        assert covered == nunits
    assert lib.wl_shard_range(10, 4, 4, C.byref(lo), C.byref(hi)) == lib.wl_shard_range(10, 0, 0, C.byref(lo), C.byref(hi)) != 0


def test_header_compiles_as_c():
    """The boundary is plain C: no C++/HIP/torch types in the signatures."""
    src = '#include "wavelets_mi355x.h"\nint main(void){ return wl_version == 0; }\n'
    p = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                       input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()


def test_no_device_is_loud(W):
    import torch
    if torch.cuda.is_available():
        return
    lib = W._lib.load()
    out = C.c_void_p()
    assert lib.wl_ctx_create(0, C.byref(out)) == -13      # WL_ENODEVICE, never a CPU fallback
    assert not out.value


def test_c_program_links_and_runs(W, tmp_path):
    """A C99 program (tests/abi_demo.c) built with plain gcc against the header and the shared library."""
    exe = str(tmp_path / "abi_demo")
    libdir = os.path.dirname(W._lib.LIB_PATH)
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_demo.c"),
                        "-L", libdir, "-lwavelets_mi355x", "-Wl,-rpath," + libdir, "-o", exe], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    r = subprocess.run([exe], capture_output=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout.decode(), r.stderr.decode())
    assert b"abi_demo:" in r.stdout
