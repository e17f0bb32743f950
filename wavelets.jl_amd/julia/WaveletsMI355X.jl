# WaveletsMI355X.jl -- Julia glue for libwavelets_mi355x.so (include/wavelets_mi355x.h).
#
# NOT executed in this repository's CI: the build image has no Julia.  It is the binding a
# Wavelets.jl maintainer adds (as a package extension next to ext/WaveletsGPUExt, wired through
# [weakdeps]/[extensions] of Project.toml with AMDGPU as the trigger).  It only adds methods to the
# reference's internal seam -- `Transforms._dwt!` / `_wpt!` (src/Transforms/transforms_main.jl:105-176
# end there) -- for `ROCArray`s, so `dwt / idwt / dwt! / idwt! / wpt / iwpt` and every caller of them
# (denoise, bestbasistree, ...) pick the MI355X backend by array-type dispatch.  A `ROCVector` method
# is more specific than the extension's `AbstractGPUVector` method, so it wins without touching it.
module WaveletsMI355X

using Wavelets
using Wavelets: WT, Util
using Wavelets.Transforms: Transforms
using AMDGPU: ROCArray, ROCVector, ROCMatrix, AMDGPU

# The library path is fixed at load time (`ccall` needs a constant).  Default: the bit-exact build.  Pointing
# WAVELETS_MI355X_LIB at libwavelets_mi355x_fma.so selects the opt-in fused arithmetic mode (same ABI, same kernels built with FMA
# contraction: agrees with the CPU reference to 1e-6*sqrt(L) relative L2 in Float32 / 1e-13*sqrt(L) in Float64, not bit for bit).
const LIB = get(ENV, "WAVELETS_MI355X_LIB", "libwavelets_mi355x.so")
const DT = Dict(Float32 => Cint(0), Float64 => Cint(1))
# One context per (device, stream): a wl_ctx owns a workspace that kernels queued on its stream are still using, so two
# tasks (= two HIP streams in AMDGPU.jl) must never share one.  Independent transforms issued from several tasks then
# overlap on the GPU (the small levels of one hide behind the first kernel of the next).
const CTX = Dict{Tuple{Int,UInt},Ptr{Cvoid}}()
const CTX_LOCK = ReentrantLock()
const OPTIONS = Dict{String,Int64}()                     # wl_ctx_set_option pairs applied to every context

"""Destroy every context (and its device workspace) created so far; new ones are created on demand.  Registered with
`atexit`, and useful after a burst of work on short-lived tasks: a context is keyed by (device, stream) and would
otherwise live as long as the process."""
function destroy_contexts!()
    lock(CTX_LOCK) do
        for h in values(CTX)
            ccall((:wl_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), h)
        end
        empty!(CTX)
    end
    return nothing
end
__init__() = atexit(destroy_contexts!)

"""`set_option!("WL_TJ", 128)`: tuning / test switch on every context (DESIGN.md section 5 lists them; none changes a result)."""
function set_option!(key::AbstractString, value::Integer)
    lock(CTX_LOCK) do
        OPTIONS[String(key)] = Int64(value)
        for h in values(CTX)
            check(ccall((:wl_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), h, key, value))
        end
    end
    return nothing
end

stream() = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))     # hipStream_t of the current task
function ctx()
    dev = AMDGPU.device_id(AMDGPU.device()) - 1
    key = (dev, UInt(stream()))
    lock(CTX_LOCK) do
        get!(CTX, key) do
            r = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:wl_ctx_create, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}), dev, r))
            for (k, v) in OPTIONS
                check(ccall((:wl_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), r[], k, v))
            end
            r[]
        end
    end
end

# status -> the exception the reference throws (transforms_filter.jl:25-34, transforms_lifting.jl:131-139)
function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:wl_strerror, LIB), Cstring, (Cint,), rc))
    rc == -4 && throw(DimensionMismatch(msg))
    rc in (-1, -2, -3, -5, -6, -7, -9, -10) && throw(ArgumentError(msg))
    error("libwavelets_mi355x: $msg (status $rc)")
end

dims3(x) = Int64[size(x)..., ntuple(_ -> 1, 3 - ndims(x))...]

# ---- filter bank: replaces _dwt!(y, x, filter::OrthoFilter, L, fw), transforms_filter.jl:13,113,192
# Dispatch: WaveletsGPUExt (loaded with AMDGPU, since AMDGPU loads GPUArrays + KernelAbstractions) defines
#   _dwt!(y::AbstractGPUVector{Ty}, x::AbstractGPUVector{Tx}, ...), ...::AbstractGPUMatrix..., ...::AbstractGPUArray{.,3}
# (ext/WaveletsGPUExt/filter_transforms_gpu.jl:171,216,271; lifting_transforms_gpu.jl:171,210,249).  The methods below mirror
# them one for one on ROCVector / ROCMatrix / ROCArray{T,3}: each signature is a strict subtype of the extension's, so it is the
# more specific method for same-T Float32/Float64 arguments by the subtype rule alone (no specificity heuristics, no ambiguity);
# mixed element types (Tx != Ty), Integer and Complex arrays do not match and fall through to the extension, which promotes.
# tests/test_julia_glue.py proves the element-wise `<:` statically against the reference's signatures.
function dwt_filter_device!(y, x, filter::OrthoFilter, L::Integer, fw::Bool, ::Type{T}, N::Int) where {T}
    size(x) == size(y) || throw(DimensionMismatch("in and out array size must match"))
    q = filter.qmf                                  # Float64 taps; converted to T inside, like makereverseqmfpair
    GC.@preserve y x check(ccall((:wl_dwt_filter, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Float64}, Cint, Cint, Cint, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), pointer(x), N, dims3(x), q, length(q), L, fw, stream()))
    return y
end
function Transforms._dwt!(y::ROCVector{T}, x::ROCVector{T}, filter::OrthoFilter, L::Integer, fw::Bool) where {T<:Union{Float32,Float64}}
    return dwt_filter_device!(y, x, filter, L, fw, T, 1)
end
function Transforms._dwt!(y::ROCMatrix{T}, x::ROCMatrix{T}, filter::OrthoFilter, L::Integer, fw::Bool) where {T<:Union{Float32,Float64}}
    return dwt_filter_device!(y, x, filter, L, fw, T, 2)
end
function Transforms._dwt!(y::ROCArray{T,3}, x::ROCArray{T,3}, filter::OrthoFilter, L::Integer, fw::Bool) where {T<:Union{Float32,Float64}}
    return dwt_filter_device!(y, x, filter, L, fw, T, 3)
end

# ---- lifting: replaces _dwt!(y, scheme::GLS, L, fw), transforms_lifting.jl:30,128,200
function flatten(s::GLS)
    isup = Int32[st.steptype isa WT.UpdateStep for st in s.step]
    nc = Int32[length(st.param.coef) for st in s.step]
    sh = Int32[st.param.shift for st in s.step]
    cf = Float64[c for st in s.step for c in st.param.coef]
    return isup, nc, sh, cf
end
function dwt_lifting_device!(y, scheme::GLS, L::Integer, fw::Bool, ::Type{T}, N::Int) where {T}
    isup, nc, sh, cf = flatten(scheme)
    GC.@preserve y check(ccall((:wl_dwt_lifting, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Int64}, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
                 Cdouble, Cdouble, Cint, Cint, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), N, dims3(y), length(isup), isup, nc, sh, cf,
                scheme.norm1, scheme.norm2, L, fw, stream()))
    return y
end
function Transforms._dwt!(y::ROCVector{T}, scheme::GLS, L::Integer, fw::Bool) where {T<:Union{Float32,Float64}}
    return dwt_lifting_device!(y, scheme, L, fw, T, 1)
end
function Transforms._dwt!(y::ROCMatrix{T}, scheme::GLS, L::Integer, fw::Bool) where {T<:Union{Float32,Float64}}
    return dwt_lifting_device!(y, scheme, L, fw, T, 2)
end
function Transforms._dwt!(y::ROCArray{T,3}, scheme::GLS, L::Integer, fw::Bool) where {T<:Union{Float32,Float64}}
    return dwt_lifting_device!(y, scheme, L, fw, T, 3)
end

# dwt(x, scheme::GLS, L) / idwt(x, scheme, L) of the reference are `y = similar(x); copyto!(y, x); _dwt!(y, scheme, L, fw)`
# (transforms_main.jl:119-124).  The library's out-of-place entry fuses the copy away (same bits, one pass less).
for (f, fw) in ((:dwt, true), (:idwt, false))
    @eval function Transforms.$f(x::ROCArray{T,N}, scheme::GLS, L::Integer=Util.maxtransformlevels(x)) where {T<:Union{Float32,Float64},N}
        y = similar(x)
        isup, nc, sh, cf = flatten(scheme)
        GC.@preserve y x check(ccall((:wl_dwt_lifting_oop, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Int64}, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
                     Cdouble, Cdouble, Cint, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(y), pointer(x), N, dims3(x), length(isup), isup, nc, sh, cf,
                    scheme.norm1, scheme.norm2, L, $fw, stream()))
        return y
    end
end

# ---- wavelet packets: replaces _wpt!, transforms_filter.jl:301, transforms_lifting.jl:283
function Transforms._wpt!(y::ROCVector{T}, x::ROCVector{T}, filter::OrthoFilter, tree::BitVector,
                          fw::Bool) where {T<:Union{Float32,Float64}}
    size(x) == size(y) || throw(DimensionMismatch("in and out array size must match"))
    t = UInt8.(tree)
    GC.@preserve y x check(ccall((:wl_wpt_filter, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Ptr{UInt8}, Int64, Cint, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), pointer(x), length(x), filter.qmf, length(filter.qmf), t, length(t), fw, stream()))
    return y
end
function Transforms._wpt!(y::ROCVector{T}, scheme::GLS, tree::BitVector, fw::Bool) where {T<:Union{Float32,Float64}}
    isup, nc, sh, cf = flatten(scheme)
    t = UInt8.(tree)
    GC.@preserve y check(ccall((:wl_wpt_lifting, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
                 Cdouble, Cdouble, Ptr{UInt8}, Int64, Cint, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), length(y), length(isup), isup, nc, sh, cf, scheme.norm1, scheme.norm2,
                t, length(t), fw, stream()))
    return y
end

# wpt(x, wt, L::Integer) / wpt!(y, x, filter, L::Integer) / wpt!(y, scheme, L::Integer) of the reference build
# maketree(length(x), L, :full) -- a BitVector of n - 1 nodes -- and walk it (transforms_main.jl:134-176).  For device
# arrays the depth is all the library needs: these methods are more specific than the reference's generic ones.
for (f, fw) in ((:wpt!, true), (:iwpt!, false))
    @eval function Transforms.$f(y::ROCVector{T}, x::ROCVector{T}, filter::OrthoFilter,
                                 L::Integer=Util.maxtransformlevels(x)) where {T<:Union{Float32,Float64}}
        size(x) == size(y) || throw(DimensionMismatch("in and out array size must match"))
        0 <= L <= Util.maxtransformlevels(x) || throw(AssertionError("0 <= L <= maxtransformlevels(n)"))   # maketree's @assert
        GC.@preserve y x check(ccall((:wl_wpt_filter_full, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Cint, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(y), pointer(x), length(x), filter.qmf, length(filter.qmf), L, $fw, stream()))
        return y
    end
    @eval function Transforms.$f(y::ROCVector{T}, scheme::GLS, L::Integer=Util.maxtransformlevels(y)) where {T<:Union{Float32,Float64}}
        0 <= L <= Util.maxtransformlevels(y) || throw(AssertionError("0 <= L <= maxtransformlevels(n)"))
        isup, nc, sh, cf = flatten(scheme)
        GC.@preserve y check(ccall((:wl_wpt_lifting_full, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
                     Cdouble, Cdouble, Cint, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(y), length(y), length(isup), isup, nc, sh, cf, scheme.norm1, scheme.norm2,
                    L, $fw, stream()))
        return y
    end
end
for (f, fb) in ((:wpt, :wpt!), (:iwpt, :iwpt!))
    @eval function Transforms.$f(x::ROCVector{T}, filter::OrthoFilter, L::Integer=Util.maxtransformlevels(x)) where {T<:Union{Float32,Float64}}
        return Transforms.$fb(similar(x), x, filter, L)
    end
    @eval function Transforms.$f(x::ROCVector{T}, scheme::GLS, L::Integer=Util.maxtransformlevels(x)) where {T<:Union{Float32,Float64}}
        return Transforms.$fb(copy(x), scheme, L)
    end
end

# ---- dwtc / idwtc: named but never defined by the reference (transforms_main.jl:179-181)
for (f, fw) in ((:dwtc, true), (:idwtc, false))
    @eval function $f(x::ROCMatrix{T}, filter::OrthoFilter, L::Integer=Util.maxtransformlevels(size(x, 1))) where {T}
        y = similar(x)
        GC.@preserve y x check(ccall((:wl_dwtc_filter, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Cint, Cint, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(y), pointer(x), size(x, 1), size(x, 2), size(x, 1),
                    filter.qmf, length(filter.qmf), L, $fw, stream()))
        return y
    end
end

for (f, fw) in ((:dwtc, true), (:idwtc, false))
    @eval function $f(x::ROCMatrix{T}, scheme::GLS, L::Integer=Util.maxtransformlevels(size(x, 1))) where {T<:Union{Float32,Float64}}
        y = similar(x)                              # out of place straight from x: no copy, no in-place staging
        isup, nc, sh, cf = flatten(scheme)
        GC.@preserve y x check(ccall((:wl_dwtc_lifting_oop, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
                     Cdouble, Cdouble, Cint, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(y), pointer(x), size(y, 1), size(y, 2), size(y, 1), length(isup), isup, nc, sh, cf,
                    scheme.norm1, scheme.norm2, L, $fw, stream()))
        return y
    end
end

# ---- a batch of independent images: x[:, :, i] -> dwt(x[:, :, i], filter, L), every level ONE launch over all images ----
# (no counterpart in the reference: its 2-D transform is per image, transforms_filter.jl:113-188)
for (f, fw) in ((:dwt_batch, true), (:idwt_batch, false))
    @eval function $f(x::ROCArray{T,3}, filter::OrthoFilter,
                      L::Integer=min(Util.maxtransformlevels(size(x, 1)), Util.maxtransformlevels(size(x, 2)))) where {T<:Union{Float32,Float64}}
        y = similar(x)
        GC.@preserve y x check(ccall((:wl_dwt_filter_batch, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Int64, Int64, Ptr{Float64}, Cint, Cint, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(y), pointer(x), Int64[size(x, 1), size(x, 2)], size(x, 3), size(x, 1) * size(x, 2),
                    filter.qmf, length(filter.qmf), L, $fw, stream()))
        return y
    end
end

# ---- multi-GPU dwtc: shard the columns of a batch over the devices of one node --------------------------------------
# The path has no exchange step (SURVEY.md 8e): columns are independent, so rank r of `world` transforms the contiguous
# column block `shard_range(nsignals, r, world)` on its own GPU with its own context; only the wavelet description (a few
# dozen Float64) is shared -- inside one Julia process that is the `filter` object itself, across processes (MPI.jl / RCCL)
# it is one broadcast of `filter.qmf`.  No signal data crosses GPUs.
"""`shard_range(nunits, rank, world)` -> 1-based `lo:hi` column range of rank `rank` (0-based), contiguous block partition
(the library's own arithmetic: `wl_shard_range`)."""
function shard_range(nunits::Integer, rank::Integer, world::Integer)
    lo, hi = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:wl_shard_range, LIB), Cint, (Int64, Cint, Cint, Ptr{Int64}, Ptr{Int64}), nunits, rank, world, lo, hi))
    return (lo[] + 1):hi[]
end

"""`dwtc_sharded(X, wt, L; devices=AMDGPU.devices())`: batched column-wise dwt of the HOST matrix `X` (len x nsignals) on
several GPUs of one node: one task per device, each uploads its column block, runs `dwtc` there (its own context and
stream) and downloads the coefficients into the matching block of the result.  A resident pipeline keeps the blocks on the
devices instead (call `dwtc` on each device's `ROCMatrix` from that device's task); this function is the documented recipe."""
function dwtc_sharded(X::AbstractMatrix{T}, wt, L::Integer=Util.maxtransformlevels(size(X, 1));
                      devices=AMDGPU.devices()) where {T<:Union{Float32,Float64}}
    Y = similar(X)
    world = length(devices)
    @sync for (r, dev) in enumerate(devices)
        Threads.@spawn begin
            AMDGPU.device!(dev)                              # this task's device (and with it this task's stream and context)
            cols = shard_range(size(X, 2), r - 1, world)
            if !isempty(cols)
                y = dwtc(ROCArray(X[:, cols]), wt, L)
                copyto!(view(Y, :, cols), Array(y))
            end
        end
    end
    return Y
end

# ---- MODWT: replaces modwt / imodwt, transforms_maximal_overlap.jl:47-63, 99-107 ----------------------
function Transforms.modwt(x::ROCVector{T}, wt::OrthoFilter, L::Integer=Util.maxmodwttransformlevels(x)) where {T<:Union{Float32,Float64}}
    L <= Util.maxmodwttransformlevels(x) || throw(ArgumentError("Too many transform levels (length(x) < 2^L)"))
    L >= 1 || throw(ArgumentError("L must be >= 1"))
    n = length(x)
    out = similar(x, n, L + 1)
    GC.@preserve out x check(ccall((:wl_modwt, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Cint, Ptr{Cvoid}),
                ctx(), DT[T], pointer(out), n, pointer(x), n, wt.qmf, length(wt.qmf), L, stream()))
    return out
end
function Transforms.imodwt(xw::ROCMatrix{T}, wt::OrthoFilter) where {T<:Union{Float32,Float64}}
    n, nc = size(xw)
    x = similar(xw, n)
    GC.@preserve x xw check(ccall((:wl_imodwt, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Float64}, Cint, Ptr{Cvoid}),
                ctx(), DT[T], pointer(x), pointer(xw), n, n, nc, wt.qmf, length(wt.qmf), stream()))
    return x
end

# ---- Threshold: replaces threshold! (threshold_main.jl:21-117), mad! (denoising.jl:103-110), and the helpers
# of the translation-invariant branch of denoise (denoising.jl:44-66).  With these methods `denoise(x::ROCArray, ...)`
# and `noisest` run unchanged: every array operation they perform dispatches here.
using Wavelets.Threshold: Threshold, HardTH, SoftTH, SemiSoftTH, SteinTH, BiggestTH, PosTH, NegTH
const THCODE = Dict(HardTH => Cint(0), SoftTH => Cint(1), SemiSoftTH => Cint(2), SteinTH => Cint(3), PosTH => Cint(4), NegTH => Cint(5))
# Julia computes `x[i] op t` in promote_type(T, typeof(t)): Float64 only when T is Float32 and t is a Float64
t_is_f64(::Type{T}, t) where {T} = Cint(promote_type(T, typeof(t)) === Float64 && T !== Float64)
# One method per threshold type, like the reference (threshold_main.jl:35,48,64,82,98,110): `th::Union{HardTH,...}` would be
# more specific than the reference's method in the first slot and less specific in the second -- an ambiguity error at the call.
for TH in (:HardTH, :SoftTH, :SemiSoftTH, :SteinTH)
    @eval function Threshold.threshold!(x::ROCArray{T}, th::$TH, t::Real) where {T<:Union{Float32,Float64}}
        @assert t >= 0
        GC.@preserve x check(ccall((:wl_threshold, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Cint, Cdouble, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(x), length(x), THCODE[$TH], Float64(t), t_is_f64(T, t), stream()))
        return x
    end
end
for TH in (:PosTH, :NegTH)
    @eval function Threshold.threshold!(x::ROCArray{T}, th::$TH) where {T<:Union{Float32,Float64}}
        GC.@preserve x check(ccall((:wl_threshold, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Cint, Cdouble, Cint, Ptr{Cvoid}),
                    ctx(), DT[T], pointer(x), length(x), THCODE[$TH], 0.0, Cint(0), stream()))
        return x
    end
end
function Threshold.threshold!(x::ROCArray{T}, ::BiggestTH, m::Int) where {T<:Union{Float32,Float64}}
    @assert m >= 0
    GC.@preserve x check(ccall((:wl_threshold_biggest, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                ctx(), DT[T], pointer(x), length(x), m, stream()))
    return x
end
function Threshold.mad!(y::ROCArray{T}) where {T<:Union{Float32,Float64}}
    r = Ref{Cdouble}(0)
    GC.@preserve y check(ccall((:wl_mad, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), length(y), r, stream()))
    return T(r[])
end
function Util.circshift!(b::ROCVector{T}, a::ROCVector{T}, shift::Integer) where {T<:Union{Float32,Float64}}
    GC.@preserve b a check(ccall((:wl_circshift, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}),
                ctx(), DT[T], pointer(b), pointer(a), 1, Int64[length(a)], Int64[shift], stream()))
    return b
end
function Threshold.arrayadd!(y::ROCArray{T}, z::ROCArray{T}) where {T<:Union{Float32,Float64}}
    length(y) == length(z) || throw(DimensionMismatch("lengths must be equal"))
    GC.@preserve y z check(ccall((:wl_arrayadd, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), pointer(z), length(y), stream()))
    return y
end


# ---- denoise, translation-invariant branch (denoising.jl:36-67) as one device-resident batch ----------------------
# The generic method of the reference already works on ROCArrays through the methods above (one spin at a time, a host
# round trip for sigma).  For orthogonal filters on vectors and square matrices this method runs all prod(nspin) spins
# as one batch and keeps sigma on the device; other argument combinations fall through to the generic method.
using Wavelets.Threshold: DNFT, VisuShrink, noisest
function Threshold.denoise(x::ROCArray{T,N}, wt::OrthoFilter=Threshold.DEFAULT_WAVELET;
                           L::Int=min(Util.maxtransformlevels(x), 6), dnt::S=VisuShrink(size(x, 1)),
                           estnoise::Function=noisest, TI::Bool=false,
                           nspin::Union{Int,Tuple}=tuple([8 for i = 1:ndims(x)]...)) where {T<:Union{Float32,Float64},N,S<:DNFT}
    # only the reference's own TI branch: threshold!(xt, th, t) exists for Hard / Soft / Semisoft / Stein (Pos / Neg take no t:
    # the generic method raises its MethodError); matrices need one nspin entry per dimension
    nspt = nspin isa Int ? (nspin,) : nspin
    if !(TI && (N == 1 || (N == 2 && length(nspt) == 2)) && get(THCODE, typeof(dnt.th), Cint(9)) <= 3)
        return invoke(Threshold.denoise, Tuple{AbstractArray,Union{Wavelets.WT.DiscreteWavelet,Nothing}}, x, wt;
                      L=L, dnt=dnt, estnoise=estnoise, TI=TI, nspin=nspin)
    end
    Util.iscube(x) || throw(ArgumentError("array must be square/cube"))
    sigma = estnoise === noisest ? -1.0 : Float64(estnoise(x, wt))
    estnoise === noisest || (sigma >= 0 && sigma * dnt.t >= 0) || throw(AssertionError("t >= 0"))   # threshold_main.jl:24 (+Inf passes, NaN does not)
    y = similar(x)
    nsp = N == 1 ? Int64[prod(nspt), 1, 1] : Int64[nspt..., 1]      # vectors: prod(nspin) spins shifted by 0 .. pns-1 (denoising.jl:38-42)
    GC.@preserve y x check(ccall((:wl_denoise_ti_filter, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Float64}, Cint, Cint, Cint, Cdouble, Ptr{Int64},
                 Cdouble, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), pointer(x), N, dims3(x), wt.qmf, length(wt.qmf), L, THCODE[typeof(dnt.th)],
                Float64(dnt.t), nsp, sigma, stream()))
    return y
end

# ... and for lifting schemes (wl_denoise_ti_lifting): same conditions, same fall-through
function Threshold.denoise(x::ROCArray{T,N}, wt::GLS;
                           L::Int=min(Util.maxtransformlevels(x), 6), dnt::S=VisuShrink(size(x, 1)),
                           estnoise::Function=noisest, TI::Bool=false,
                           nspin::Union{Int,Tuple}=tuple([8 for i = 1:ndims(x)]...)) where {T<:Union{Float32,Float64},N,S<:DNFT}
    nspt = nspin isa Int ? (nspin,) : nspin
    if !(TI && (N == 1 || (N == 2 && length(nspt) == 2)) && get(THCODE, typeof(dnt.th), Cint(9)) <= 3)
        return invoke(Threshold.denoise, Tuple{AbstractArray,Union{Wavelets.WT.DiscreteWavelet,Nothing}}, x, wt;
                      L=L, dnt=dnt, estnoise=estnoise, TI=TI, nspin=nspin)
    end
    Util.iscube(x) || throw(ArgumentError("array must be square/cube"))
    sigma = estnoise === noisest ? -1.0 : Float64(estnoise(x, wt))
    estnoise === noisest || (sigma >= 0 && sigma * dnt.t >= 0) || throw(AssertionError("t >= 0"))
    y = similar(x)
    isup, nc, sh, cf = flatten(wt)
    nsp = N == 1 ? Int64[prod(nspt), 1, 1] : Int64[nspt..., 1]
    GC.@preserve y x check(ccall((:wl_denoise_ti_lifting, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Int64}, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
                 Cdouble, Cdouble, Cint, Cint, Cdouble, Ptr{Int64}, Cdouble, Ptr{Cvoid}),
                ctx(), DT[T], pointer(y), pointer(x), N, dims3(x), length(isup), isup, nc, sh, cf, wt.norm1, wt.norm2,
                L, THCODE[typeof(dnt.th)], Float64(dnt.t), nsp, sigma, stream()))
    return y
end

end # module
