"""Lint of the Julia glue (wavelets.jl_amd/julia/WaveletsMI355X.jl) against the C ABI -- CPU only.

The image has no Julia, so the 22 `ccall`s of the glue have never run; one wrong `Cint` / `Int64` is a segfault on a
maintainer's first call.  This test parses every `ccall((:wl_x, LIB), Ret, (types...), args...)`, maps the Julia C types to
ctypes and compares name, return type, arity and every argument type with `_lib.SIGNATURES` (which tests/test_abi.py pins to
include/wavelets_mi355x.h and to `nm -D` of the built library).  It also checks that every data-path entry point the glue is
supposed to bind is bound, that the status codes the glue maps to Julia exceptions are the header's, and that every call
handing a device pointer to C keeps its arrays alive with `GC.@preserve`.
Seam mirrored by the glue: /root/reference/src/Transforms/transforms_main.jl:105-176, ext/WaveletsGPUExt/filter_transforms_gpu.jl:171-187.
"""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "wavelets.jl_amd", "julia", "WaveletsMI355X.jl")
HEADER = os.path.join(ROOT, "include", "wavelets_mi355x.h")

JL2C = {
    "Cint": C.c_int, "Int64": C.c_int64, "Cdouble": C.c_double, "Float64": C.c_double, "Cstring": C.c_char_p,
    "Csize_t": C.c_size_t, "Ptr{Cvoid}": C.c_void_p, "Ptr{Ptr{Cvoid}}": C.POINTER(C.c_void_p),
    "Ptr{Int64}": C.POINTER(C.c_int64), "Ptr{Int32}": C.POINTER(C.c_int32), "Ptr{UInt8}": C.POINTER(C.c_uint8),
    "Ptr{Float64}": C.POINTER(C.c_double), "Ptr{Cdouble}": C.POINTER(C.c_double),
}


def _split_top(s):
    """split at top-level commas (parentheses, brackets and braces nest)"""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        out.append("".join(cur).strip())
    return out


def _ccalls():
    src = open(GLUE).read()
    src = re.sub(r"#[^\n]*", "", src)            # comments (no '#' inside the glue's string literals on ccall lines)
    calls = []
    for m in re.finditer(r"ccall\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        parts = _split_top(src[i:j - 1])
        sym = re.match(r"\(:(\w+),\s*LIB\)", parts[0]).group(1)
        ret = parts[1]
        types = _split_top(parts[2].strip()[1:-1])
        args = parts[3:]
        line = src.count("\n", 0, m.start()) + 1
        # the statement's line prefix (for the GC.@preserve check)
        ls = src.rfind("\n", 0, m.start()) + 1
        calls.append({"sym": sym, "ret": ret, "types": types, "args": args, "line": line, "prefix": src[ls:m.start()]})
    return calls


def test_every_ccall_matches_the_abi():
    from wavelets_jl_amd import _lib
    calls = _ccalls()
    assert len(calls) >= 22
    for c in calls:
        assert c["sym"] in _lib.SIGNATURES, "ccall of an unknown symbol %s (line %d)" % (c["sym"], c["line"])
        restype, argtypes = _lib.SIGNATURES[c["sym"]]
        assert JL2C[c["ret"]] is restype, (c["sym"], c["line"], c["ret"], restype)
        assert len(c["types"]) == len(argtypes), "%s (line %d): %d argument types, the ABI has %d" % (c["sym"], c["line"], len(c["types"]), len(argtypes))
        assert len(c["args"]) == len(c["types"]), "%s (line %d): %d values for %d argument types" % (c["sym"], c["line"], len(c["args"]), len(c["types"]))
        for k, (jt, ct) in enumerate(zip(c["types"], argtypes)):
            assert jt in JL2C, (c["sym"], c["line"], jt)
            got = JL2C[jt]
            # ctypes caches POINTER(T): identical types are the same object
            assert got is ct, "%s (line %d) argument %d: Julia %s, ABI %s" % (c["sym"], c["line"], k + 1, jt, ct)


def test_glue_binds_every_data_path_entry_point():
    used = {c["sym"] for c in _ccalls()}
    required = {
        "wl_ctx_create", "wl_ctx_destroy", "wl_ctx_set_option", "wl_strerror", "wl_shard_range",
        "wl_dwt_filter", "wl_dwt_lifting", "wl_dwt_lifting_oop", "wl_wpt_filter", "wl_wpt_lifting", "wl_wpt_filter_full", "wl_wpt_lifting_full",
        "wl_dwtc_filter", "wl_dwtc_lifting_oop", "wl_dwt_filter_batch", "wl_modwt", "wl_imodwt",
        "wl_threshold", "wl_threshold_biggest", "wl_mad", "wl_circshift", "wl_arrayadd", "wl_denoise_ti_filter", "wl_denoise_ti_lifting",
    }
    assert required <= used, sorted(required - used)


def test_device_pointers_are_gc_preserved():
    for c in _ccalls():
        names = []
        for a in c["args"]:
            for m in re.finditer(r"pointer\((\w+)\)", a):
                names.append(m.group(1))
        if not names:
            continue
        m = re.search(r"GC\.@preserve\s+([\w\s]+?)\s+check\($", c["prefix"].rstrip() + "")
        assert m, "%s (line %d): pointer(...) passed to C outside GC.@preserve" % (c["sym"], c["line"])
        kept = set(m.group(1).split())
        assert set(names) <= kept, (c["sym"], c["line"], names, kept)


def test_status_codes_mapped_by_the_glue_are_the_headers():
    hdr = open(HEADER).read()
    codes = {name: int(val) for name, val in re.findall(r"#define\s+(WL_E\w+)\s+\(?(-?\d+)\)?", hdr)}
    if not codes:
        codes = {name: int(val) for name, val in re.findall(r"(WL_E\w+)\s*=\s*(-?\d+)", hdr)}
    assert codes.get("WL_EDIMS") == -4                       # -> DimensionMismatch in check()
    arg_errors = {codes[k] for k in ("WL_EINVAL_SIZE", "WL_EINVAL_L", "WL_EALIAS", "WL_EINVAL_CUBE", "WL_EINVAL_TREE", "WL_EINVAL_SCHEME",
                                     "WL_EINVAL_FILTER", "WL_EINVAL_ARG")}
    src = open(GLUE).read()
    m = re.search(r"rc in \(([^)]*)\) && throw\(ArgumentError", src)
    assert {int(v) for v in m.group(1).split(",")} == arg_errors
    assert "rc == -4 && throw(DimensionMismatch" in src
