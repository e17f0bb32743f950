# runtests_mi355x.jl -- what a maintainer runs on an MI355X box with Julia, AMDGPU.jl and Wavelets.jl installed:
#
#     WAVELETS_MI355X_LIB=/path/to/libwavelets_mi355x.so julia --project runtests_mi355x.jl
#
# It mirrors the reference's GPU test file (test/gpu.jl:14-86, 169-183): the same wavelets, sizes and depths, ROCArray
# inputs against the CPU path of Wavelets.jl itself -- but with tolerance ZERO where the reference allows 1e-6 / 1e-5,
# because this backend evaluates the reference's sums in the reference's order without FMA contraction.  (NOT run in this
# repository: the build image has no Julia.  The same comparisons run against the C restatement of the reference in
# tests/test_gpu_parity.py.)
using Test
using Wavelets
using AMDGPU
include(joinpath(@__DIR__, "WaveletsMI355X.jl"))
using .WaveletsMI355X

same(a, b) = Array(b) == a          # bit-for-bit (0.0 == -0.0)

@testset "MI355X backend vs Wavelets.jl CPU path" begin
    @testset "1-D filter ($T)" for T in (Float32, Float64)
        for wname in (WT.haar, WT.db2, WT.db4, WT.sym5, WT.coif4), n in (16, 64, 4096, 1 << 20)
            wt = wavelet(wname)
            x = randn(T, n)
            xg = ROCArray(x)
            for L in unique((1, min(3, maxtransformlevels(n)), maxtransformlevels(n)))
                y = dwt(x, wt, L)
                yg = dwt(xg, wt, L)
                @test same(y, yg)
                @test same(idwt(y, wt, L), idwt(yg, wt, L))
            end
        end
    end
    @testset "1-D lifting ($T)" for T in (Float32, Float64)
        for wname in (WT.haar, WT.db2, WT.cdf97), n in (16, 64, 4096, 1 << 20)
            wt = wavelet(wname, WT.Lifting)
            x = randn(T, n)
            for L in unique((1, min(3, maxtransformlevels(n)), maxtransformlevels(n)))
                y = dwt(x, wt, L)
                yg = dwt(ROCArray(x), wt, L)            # out-of-place method (wl_dwt_lifting_oop)
                @test same(y, yg)
                z = ROCArray(x); dwt!(z, wt, L)         # in place
                @test same(y, z)
                @test same(idwt(y, wt, L), idwt(yg, wt, L))
            end
        end
    end
    @testset "2-D / 3-D filter" begin
        for wname in (WT.haar, WT.db4), n in (8, 16, 512)
            wt = wavelet(wname)
            x = randn(Float32, n, n)
            for L in 1:min(2, maxtransformlevels(n))
                y = dwt(x, wt, L)
                @test same(y, dwt(ROCArray(x), wt, L))
                @test same(idwt(y, wt, L), idwt(ROCArray(y), wt, L))
            end
        end
        x = randn(Float32, 8, 8, 8); wt = wavelet(WT.haar)
        for L in 1:2
            @test same(dwt(x, wt, L), dwt(ROCArray(x), wt, L))
        end
        x = randn(Float32, 8192, 8192); wt = wavelet(WT.db4)    # the headline configuration
        @test same(dwt(x, wt), dwt(ROCArray(x), wt))
    end
    @testset "2-D lifting" begin
        for wname in (WT.haar, WT.db2, WT.cdf97), n in (8, 16, 1024)
            wt = wavelet(wname, WT.Lifting)
            x = randn(Float32, n, n)
            for L in 1:min(2, maxtransformlevels(n))
                @test same(dwt(x, wt, L), dwt(ROCArray(x), wt, L))
            end
        end
    end
    @testset "wavelet packets" begin
        x = randn(Float64, 1024); wt = wavelet(WT.db2)
        for tree in (maketree(1024, 4, :full), maketree(1024, 6, :dwt))
            y = wpt(x, wt, tree)
            @test same(y, wpt(ROCArray(x), wt, tree))
            @test same(iwpt(y, wt, tree), iwpt(ROCArray(y), wt, tree))
        end
    end
    @testset "argument contract" begin
        x = ROCArray(randn(Float64, 24)); wt = wavelet(WT.db2)
        @test_throws ArgumentError dwt(x, wt, 4)
        @test_throws ArgumentError dwt(x, wt, -1)
        @test_throws ArgumentError dwt!(x, x, wt, 1)
        @test_throws DimensionMismatch dwt!(similar(x, 12), x, wt, 1)
        @test_throws ArgumentError dwt(ROCArray(randn(8, 16)), wavelet(WT.db2, WT.Lifting), 1)
    end
    @testset "denoise" begin
        n = 2^11
        x0 = testfunction(n, "Doppler")
        x = x0 + 0.05 * randn(n)
        for kw in ((TI=false,), (TI=true,), (TI=true, nspin=8))
            @test same(denoise(x; kw...), denoise(ROCArray(x); kw...))
        end
        a = randn(Float32, 64, 64)
        @test same(denoise(a, TI=true), denoise(ROCArray(a), TI=true))
    end
    @testset "modwt" begin
        x = randn(Float32, 1000); wt = wavelet(WT.db4)
        w = modwt(x, wt, 4)
        @test same(w, modwt(ROCArray(x), wt, 4))
        @test same(imodwt(w, wt), imodwt(ROCArray(w), wt))
    end
end
WaveletsMI355X.destroy_contexts!()
