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


# ---- dispatch: every method the glue adds to the reference's functions is element-wise `<:` one of the reference's own ----
# (tests/golden/reference_seam_signatures.json <- tools/gen_seam_signatures.py run against /root/reference; data, not source).
# Julia picks the more specific of two applicable methods by the subtype relation of their signatures; when neither signature is
# a subtype of the other it falls back on heuristics or reports an ambiguity.  The glue therefore only adds methods whose whole
# parameter tuple is a subtype of an existing method's (same function, same arity): then "more specific" holds by construction.
import json

SEAM = os.path.join(ROOT, "tests", "golden", "reference_seam_signatures.json")
ARRAY_ALIASES = {       # alias -> (family, fixed rank or None)
    "ROCVector": ("ROCArray", 1), "ROCMatrix": ("ROCArray", 2), "ROCArray": ("ROCArray", None),
    "AbstractGPUVector": ("AbstractGPUArray", 1), "AbstractGPUMatrix": ("AbstractGPUArray", 2), "AbstractGPUArray": ("AbstractGPUArray", None),
    "AbstractVector": ("AbstractArray", 1), "AbstractMatrix": ("AbstractArray", 2), "AbstractArray": ("AbstractArray", None),
    "Vector": ("Array", 1), "Matrix": ("Array", 2), "Array": ("Array", None),
}
ARRAY_SUPERS = {"ROCArray": {"ROCArray", "AbstractGPUArray", "AbstractArray"}, "AbstractGPUArray": {"AbstractGPUArray", "AbstractArray"},
                "AbstractArray": {"AbstractArray"}, "Array": {"Array", "AbstractArray"}}
# scalar / descriptor types: name -> set of supertypes (incl. itself)
SCALAR_SUPERS = {
    "Float32": {"Float32", "AbstractFloat", "Real", "Number", "Any"}, "Float64": {"Float64", "AbstractFloat", "Real", "Number", "Any"},
    "Bool": {"Bool", "Integer", "Real", "Number", "Any"}, "Int": {"Int", "Integer", "Real", "Number", "Any"},
    "Integer": {"Integer", "Real", "Number", "Any"}, "Real": {"Real", "Number", "Any"},
    "OrthoFilter": {"OrthoFilter", "DiscreteWavelet", "Any"}, "GLS": {"GLS", "DiscreteWavelet", "Any"},
    "BitVector": {"BitVector", "Any"},
}
for _th in ("HardTH", "SoftTH", "SemiSoftTH", "SteinTH", "BiggestTH", "PosTH", "NegTH"):
    SCALAR_SUPERS[_th] = {_th, "THType", "Any"}


def _parse_array(t, aliases):
    """'ROCArray{T,3}' -> (family, eltype expr or None, rank or None); None if t is not an array type"""
    t = t.replace(" ", "")
    m = re.match(r"^(\w+)(?:\{(.*)\})?$", t)
    if not m:
        return None
    name = aliases.get(m.group(1), m.group(1))
    if name not in ARRAY_ALIASES:
        return None
    fam, rank = ARRAY_ALIASES[name]
    args = _split_top(m.group(2)) if m.group(2) else []
    elt = args[0] if args else None
    if rank is None and len(args) > 1:
        rank = int(args[1]) if args[1].isdigit() else None
    return fam, elt, rank


def _elt_members(e, where, aliases):
    """the set of concrete leaf types an eltype expression admits, or a bound name: returns ('leaves', set) / ('bound', name)"""
    if e is None:
        return ("bound", "Any")
    if e.startswith("<:"):
        e = e[2:]
    elif e in where:
        e = where[e]
    e = aliases.get(e, e)
    if e.startswith("Union{"):
        return ("leaves", set(_split_top(e[6:-1])))
    return ("bound", e)


def _scalar_sub(a, b, aliases):
    a, b = aliases.get(a, a), aliases.get(b, b)
    if b.startswith("Union{"):
        return any(_scalar_sub(a, m, aliases) for m in _split_top(b[6:-1]))
    if a.startswith("Union{"):
        return all(_scalar_sub(m, b, aliases) for m in _split_top(a[6:-1]))
    return b in SCALAR_SUPERS.get(a, {a, "Any"})


def _param_sub(gt, gwhere, rt, rwhere, aliases):
    ga, ra = _parse_array(gt, aliases), _parse_array(rt, aliases)
    if (ga is None) != (ra is None):
        return rt == "Any"
    if ga is None:
        return _scalar_sub(gt.replace(" ", ""), rt.replace(" ", ""), aliases)
    gfam, gelt, grank = ga
    rfam, relt, rrank = ra
    if rfam not in ARRAY_SUPERS[gfam]:
        return False
    if rrank is not None and grank != rrank:
        return False
    kind, val = _elt_members(gelt, gwhere, aliases)
    rkind, rval = _elt_members(relt, rwhere, aliases)
    leaves = val if kind == "leaves" else {val}
    if rkind == "leaves":           # e.g. ValueType = Union{AbstractFloat, Complex}
        return all(any(_scalar_sub(x, m, aliases) for m in rval) for x in leaves)
    return all(_scalar_sub(x, rval, aliases) for x in leaves)


def _typevars(t):
    return set(re.findall(r"\b([A-Z][a-z]?)\b(?![{\w])", t.replace("<:", " ")))


def _method_sub(g, r, aliases):
    if len(g["params"]) != len(r["params"]):
        return False
    if not all(_param_sub(gp, g["where"], rp, r["where"], aliases) for gp, rp in zip(g["params"], r["params"])):
        return False
    # diagonal rule: a type variable the reference uses in two slots forces equal parameters there -- the glue must share one too
    for v in r["where"]:
        slots = [k for k, rp in enumerate(r["params"]) if re.search(r"[{,\s]%s[},\s]" % v, rp)]
        if len(slots) > 1:
            gv = [set(re.findall(r"[{,]\s*(\w+)\s*[},]", g["params"][k])) & set(g["where"]) for k in slots]
            if not set.intersection(*gv):
                return False
    return True


def _glue_methods(src=None):
    """methods the glue adds to reference functions (qualified Transforms. / Threshold. / Util. names), @eval loops expanded,
    one entry per arity a default argument creates"""
    src = open(GLUE).read() if src is None else src
    src = re.sub(r"#[^\n]*", lambda m: " " * len(m.group(0)), src)
    out = []
    for m in re.finditer(r"^[ \t]*(?:@eval\s+)?function\s+(Transforms|Threshold|Util)\.(\$?\w+!?)\(", src, re.M):
        i = m.end() - 1
        depth, j = 0, i
        while True:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
            if depth == 0:
                break
        wm = re.match(r"\s*where\s+\{([^\n]*)\}\s*\n", src[j:j + 300])
        where = {}
        for w in _split_top(wm.group(1)) if wm else []:
            if "<:" in w:
                v, b = [p.strip() for p in w.split("<:", 1)]
                where[v] = b.replace(" ", "")
            else:
                where[w.strip()] = "Any"
        names = [m.group(2)]
        binds = [{}]
        if "$" in src[m.start():j]:
            # innermost enclosing `for VAR in (...)` / `for (a, b) in ((..), ..)` loop at column 0
            heads = [h for h in re.finditer(r"^for (\(?[\w, ]+\)?) in \(", src, re.M) if h.start() < m.start()]
            h = heads[-1]
            d2, e = 0, h.end() - 1
            while True:
                d2 += {"(": 1, ")": -1}.get(src[e], 0)
                e += 1
                if d2 == 0:
                    break
            vars_ = [v.strip() for v in h.group(1).strip("()").split(",")]
            binds = []
            for tup in _split_top(src[h.end():e - 1]):
                vals = [v.strip().lstrip(":") for v in (_split_top(tup.strip()[1:-1]) if tup.strip().startswith("(") else [tup])]
                binds.append(dict(zip(vars_, vals)))
        positional = _split_top(_split_semicolon(src[i + 1:j - 1]))
        line = src.count("\n", 0, m.start()) + 1
        for b in binds:
            name = names[0]
            params, first_default = [], None
            for k, a in enumerate(positional):
                a = " ".join(a.split())
                parts = a.split("=", 1)
                if len(parts) > 1 and first_default is None:
                    first_default = k
                t = parts[0].split("::", 1)[1].strip() if "::" in parts[0] else "Any"
                for var, val in b.items():
                    t = t.replace("$" + var, val)
                params.append(t)
            for var, val in b.items():
                name = name.replace("$" + var, val)
            lo = len(params) if first_default is None else first_default
            for n in range(lo, len(params) + 1):
                out.append({"name": name, "params": params[:n], "where": where, "line": line})
    return out


def _split_semicolon(args):
    depth = 0
    for k, ch in enumerate(args):
        depth += {"(": 1, "[": 1, "{": 1, ")": -1, "]": -1, "}": -1}.get(ch, 0)
        if ch == ";" and depth == 0:
            return args[:k]
    return args


def _covered(g, seam):
    al = seam["aliases"]
    return [r["at"] for r in seam["methods"].get(g["name"], []) if _method_sub(g, r, al)]


def test_every_glue_method_is_a_subtype_of_a_reference_method():
    seam = json.load(open(SEAM))
    ms = _glue_methods()
    assert len(ms) >= 40, len(ms)
    names = {m["name"] for m in ms}
    assert {"_dwt!", "_wpt!", "dwt", "idwt", "wpt", "iwpt", "wpt!", "iwpt!", "modwt", "imodwt", "threshold!", "mad!", "circshift!",
            "arrayadd!", "denoise"} <= names, names
    for g in ms:
        at = _covered(g, seam)
        assert at, "glue method %s(%s) [line %d] is not element-wise <: any reference method of that name and arity" % (
            g["name"], ", ".join(g["params"]), g["line"])
    # the six seam methods mirror the in-package GPU extension one for one (rank by rank)
    want = {("_dwt!", 5): {"ext/WaveletsGPUExt/filter_transforms_gpu.jl:171", "ext/WaveletsGPUExt/filter_transforms_gpu.jl:216",
                            "ext/WaveletsGPUExt/filter_transforms_gpu.jl:271"},
            ("_dwt!", 4): {"ext/WaveletsGPUExt/lifting_transforms_gpu.jl:171", "ext/WaveletsGPUExt/lifting_transforms_gpu.jl:210",
                            "ext/WaveletsGPUExt/lifting_transforms_gpu.jl:249"}}
    for (nm, ar), ats in want.items():
        got = set()
        for g in ms:
            if g["name"] == nm and len(g["params"]) == ar:
                got |= {a for a in _covered(g, seam) if a.startswith("ext/")}
        assert got == ats, (nm, ar, got)


def test_dispatch_lint_rejects_the_round4_signatures():
    """the lint is not vacuous: the round-4 methods (`ROCArray{T,N} where N` against the extension's per-rank methods, and a Union
    of threshold types against the reference's per-type methods) are neither subtypes nor supertypes of the methods they compete
    with -- exactly the cases Julia would resolve by heuristics or report as ambiguous"""
    seam = json.load(open(SEAM))
    old = '''
function Transforms._dwt!(y::ROCArray{T,N}, x::ROCArray{T,N}, filter::OrthoFilter, L::Integer,
                          fw::Bool) where {T<:Union{Float32,Float64},N}
end
function Transforms._dwt!(y::ROCArray{T,N}, scheme::GLS, L::Integer, fw::Bool) where {T<:Union{Float32,Float64},N}
end
function Threshold.threshold!(x::ROCArray{T}, th::Union{HardTH,SoftTH,SemiSoftTH,SteinTH}, t::Real) where {T<:Union{Float32,Float64}}
end
function Threshold.threshold!(x::ROCArray{T}, th::Union{PosTH,NegTH}) where {T<:Union{Float32,Float64}}
end
'''
    ms = _glue_methods(old)
    assert len(ms) == 4
    for g in ms:
        ext_or_typed = [a for a in _covered(g, seam)]
        if g["name"] == "_dwt!":
            assert not any(a.startswith("ext/") for a in ext_or_typed), g      # not <: any of the extension's per-rank methods
        else:
            assert not ext_or_typed, g


def test_seam_fixture_is_current():
    """in the build container (where /root/reference exists) the committed fixture equals a fresh extraction"""
    import importlib.util
    import pytest
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("the reference tree is not on this machine; the committed fixture is the record")
    spec = importlib.util.spec_from_file_location("gen_seam", os.path.join(ROOT, "tools", "gen_seam_signatures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = {}
    for f in mod.FILES:
        for me in mod.methods_of(f):
            fresh.setdefault(me["name"], [])
            if me not in fresh[me["name"]]:
                fresh[me["name"]].append(me)
    assert fresh == json.load(open(SEAM))["methods"]
