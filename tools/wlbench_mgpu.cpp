// wlbench_mgpu -- the NATIVE (torch-free) multi-GPU host for the batched column-wise transform (BASELINE.json configs[4];
// SURVEY.md section 2 "RCCL bootstrap for the 8-GPU batched case", section 8e): ONE process, one host thread + one wl_ctx per
// device, RCCL over xGMI for the only collectives the path has.
//
//   wlbench_mgpu [gpus=N] [signals=65536] [len=65536] [L=16] [steps=20] [warmup=5] [filt=db4] [dry=1]
//
//   * ncclCommInitAll over the first N devices;
//   * rank 0's wavelet description (256 doubles: the packing of wavelets.jl_amd/sharding.py) reaches the other ranks by ONE
//     ncclBroadcast -- every rank builds its taps from what it RECEIVED;
//   * rank r owns the contiguous column block wl_shard_range(signals, r, N) of the len x signals batch (generated on its
//     device; no signal data ever crosses GPUs), reserves its workspace and runs wl_dwtc_filter on it;
//   * timing as bench.py: barrier, warm-up steps, synchronise, K timed steps, synchronise, barrier; the figure is the MAX over
//     ranks of the wall time of the K steps;
//   * a checksum of every shard (device reduction) is summed over ranks by one ncclAllReduce -- the cross-rank correctness
//     token -- and the MAX of the times by another.
//   Prints ONE JSON line with the keys of `bench.py --gpus N` (metric, value, unit, n_gpus, steps, warmup, ms_per_step, ...).
//   dry=1: no device is touched -- prints the partition (wl_shard_range) and the packed wavelet; used by the CPU test that
//   builds and links this file against librccl and libwavelets_mi355x.
// Not product code: a harness over the C ABI, like wlbench.cpp.  The reference has no dwtc at all
// (/root/reference/src/Transforms/transforms_main.jl:179-181).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/wavelets_mi355x.h"

static const std::map<std::string, std::vector<double>> kTaps = {
    {"haar", {0.7071067811865476, 0.7071067811865476}},
    {"db2", {0.4829629131445342, 0.8365163037378079, 0.2241438680420133, -0.12940952255126045}},
    {"db4", {0.23037781330889648, 0.7148465705529157, 0.6308807679298589, -0.027983769416860003, -0.18703481171909309, 0.030841381835560722,
             0.03288301166688518, -0.010597401785069035}},
};

constexpr int kPack = 256;          // doubles: [0] kind (0 = orthogonal filter), [1] number of taps, [2 ...] the taps

__global__ void k_fill(float *p, size_t n, unsigned long long seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (float)((double)(z >> 11) * (1.0 / 9007199254740992.0)) - 0.5f;
    }
}
// per-block partial sums in double, then one block adds them: a deterministic checksum of a shard
__global__ void k_sum(const float *p, size_t n, double *partial)
{
    __shared__ double sh[256];
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (double)p[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void k_sum_final(const double *partial, int nb, double *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += partial[i];
        out[0] = s;
    }
}

struct Barrier {                     // host threads meet here (C++17: no std::barrier)
    std::mutex m; std::condition_variable cv; int n, count = 0, gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const int g = gen;
        if (++count == n) { count = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct RankResult { int rc = 0; std::string err; double seconds = 0, checksum_all = 0, seconds_max = 0; long long lo = 0, hi = 0; std::string kernel; int taps_received = 0; };

#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { res.rc = 2; res.err = std::string("hip: ") + hipGetErrorString(e__) + " (" #x ")"; return; } } while (0)
#define NCCLCHK(x) do { ncclResult_t r__ = (x); if (r__ != ncclSuccess) { res.rc = 3; res.err = std::string("rccl: ") + ncclGetErrorString(r__) + " (" #x ")"; return; } } while (0)

static void rank_main(int rank, int world, ncclComm_t comm, Barrier &bar, const std::vector<double> &qmf0, long long signals, long long len, int L,
                      int steps, int warmup, RankResult &res)
{
    HIPCHK(hipSetDevice(rank));
    hipStream_t st;
    HIPCHK(hipStreamCreate(&st));
    // ---- the one data-path-adjacent collective: the wavelet description, from rank 0 ----
    double *dpack = nullptr;
    HIPCHK(hipMalloc(&dpack, kPack * sizeof(double)));
    std::vector<double> pack(kPack, 0.0);
    if (rank == 0) {
        pack[0] = 0.0; pack[1] = (double)qmf0.size();
        for (size_t i = 0; i < qmf0.size(); ++i) pack[2 + i] = qmf0[i];
    }
    HIPCHK(hipMemcpyAsync(dpack, pack.data(), kPack * sizeof(double), hipMemcpyHostToDevice, st));
    NCCLCHK(ncclBroadcast(dpack, dpack, kPack, ncclDouble, 0, comm, st));
    HIPCHK(hipMemcpyAsync(pack.data(), dpack, kPack * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int flen = (int)pack[1];
    if (pack[0] != 0.0 || flen < 2 || flen > 64) { res.rc = 4; res.err = "bad wavelet pack received"; return; }
    std::vector<double> qmf(pack.begin() + 2, pack.begin() + 2 + flen);      // what THIS rank received
    res.taps_received = flen;

    // ---- this rank's shard ----
    int64_t lo = 0, hi = 0;
    int rc = wl_shard_range(signals, rank, world, &lo, &hi);
    if (rc) { res.rc = 5; res.err = wl_strerror(rc); return; }
    res.lo = lo; res.hi = hi;
    const int64_t ncol = hi - lo;
    const size_t N = (size_t)ncol * (size_t)len;
    wl_ctx *ctx = nullptr;
    rc = wl_ctx_create(rank, &ctx);
    if (rc) { res.rc = 5; res.err = std::string("wl_ctx_create: ") + wl_strerror(rc); return; }
    float *x = nullptr, *y = nullptr;
    double *partial = nullptr, *dsum = nullptr, *dmax = nullptr;
    if (N) { HIPCHK(hipMalloc(&x, N * 4)); HIPCHK(hipMalloc(&y, N * 4)); }
    HIPCHK(hipMalloc(&partial, 1024 * sizeof(double)));
    HIPCHK(hipMalloc(&dsum, sizeof(double)));
    HIPCHK(hipMalloc(&dmax, sizeof(double)));
    if (N) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, x, N, 4242ull + 1000ull * rank);
    int64_t dims1[3] = {(int64_t)N, 1, 1};
    rc = wl_ctx_reserve(ctx, wl_workspace_bytes(WL_F32, 1, dims1, L));
    if (rc) { res.rc = 5; res.err = std::string("reserve: ") + wl_strerror(rc); return; }
    auto step = [&]() { return ncol ? wl_dwtc_filter(ctx, WL_F32, y, x, len, ncol, len, qmf.data(), flen, L, 1, st) : 0; };
    HIPCHK(hipStreamSynchronize(st));
    bar.wait();                                       // every rank starts its warm-up (and with it the timed steps) together
    for (int i = 0; i < warmup; ++i) { rc = step(); if (rc) { res.rc = 6; res.err = wl_strerror(rc); return; } }
    HIPCHK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) step();
    HIPCHK(hipStreamSynchronize(st));
    res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bar.wait();
    res.kernel = wl_last_kernel(ctx);
    // ---- checksum of the shard, summed over ranks; MAX of the times ----
    if (N) hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, st, y, N, partial);
    else HIPCHK(hipMemsetAsync(partial, 0, 1024 * sizeof(double), st));
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64), 0, st, partial, 1024, dsum);
    HIPCHK(hipMemcpyAsync(dmax, &res.seconds, sizeof(double), hipMemcpyHostToDevice, st));
    NCCLCHK(ncclAllReduce(dsum, dsum, 1, ncclDouble, ncclSum, comm, st));
    NCCLCHK(ncclAllReduce(dmax, dmax, 1, ncclDouble, ncclMax, comm, st));
    HIPCHK(hipMemcpyAsync(&res.checksum_all, dsum, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&res.seconds_max, dmax, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    wl_ctx_destroy(ctx);
    if (x) (void)hipFree(x);
    if (y) (void)hipFree(y);
    (void)hipFree(partial); (void)hipFree(dsum); (void)hipFree(dmax); (void)hipFree(dpack);
    (void)hipStreamDestroy(st);
}

int main(int argc, char **argv)
{
    std::map<std::string, std::string> kv = {{"gpus", "1"}, {"signals", "65536"}, {"len", "65536"}, {"L", "16"}, {"steps", "20"}, {"warmup", "5"},
                                             {"filt", "db4"}, {"dry", "0"}};
    for (int i = 1; i < argc; ++i) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { fprintf(stderr, "bad argument %s (key=value)\n", argv[i]); return 1; }
        kv[std::string(argv[i], eq - argv[i])] = eq + 1;
    }
    const int world = atoi(kv["gpus"].c_str()), L = atoi(kv["L"].c_str()), steps = atoi(kv["steps"].c_str()), warmup = atoi(kv["warmup"].c_str());
    const long long signals = atoll(kv["signals"].c_str()), len = atoll(kv["len"].c_str());
    auto it = kTaps.find(kv["filt"]);
    if (it == kTaps.end() || world < 1 || signals < 1 || len < 2 || steps < 1) { fprintf(stderr, "bad arguments\n"); return 1; }
    if (atoi(kv["dry"].c_str())) {
        // host-only: the partition and the broadcast payload (no HIP / RCCL call)
        printf("{\"dry\": true, \"n_gpus\": %d, \"wl_version\": %d, \"shards\": [", world, wl_version());
        long long covered = 0;
        for (int r = 0; r < world; ++r) {
            int64_t lo, hi;
            if (wl_shard_range(signals, r, world, &lo, &hi)) return 2;
            covered += hi - lo;
            printf("%s[%lld, %lld]", r ? ", " : "", (long long)lo, (long long)hi);
        }
        printf("], \"signals_covered\": %lld, \"pack_len\": %d, \"taps\": %d}\n", covered, kPack, (int)it->second.size());
        return 0;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < world) { fprintf(stderr, "%d GPUs requested, %d visible\n", world, ndev); return 2; }
    std::vector<ncclComm_t> comms(world);
    std::vector<int> devs(world);
    for (int r = 0; r < world; ++r) devs[r] = r;
    ncclResult_t nr = ncclCommInitAll(comms.data(), world, devs.data());
    if (nr != ncclSuccess) { fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(nr)); return 3; }
    int rccl_version = 0;
    (void)ncclGetVersion(&rccl_version);
    Barrier bar(world);
    std::vector<RankResult> res(world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back(rank_main, r, world, comms[r], std::ref(bar), std::cref(it->second), signals, len, L, steps, warmup, std::ref(res[r]));
    for (auto &t : th) t.join();
    for (int r = 0; r < world; ++r) ncclCommDestroy(comms[r]);
    long long covered = 0;
    for (int r = 0; r < world; ++r) {
        if (res[r].rc) { fprintf(stderr, "rank %d: %s\n", r, res[r].err.c_str()); return res[r].rc; }
        covered += res[r].hi - res[r].lo;
    }
    const double dt = res[0].seconds_max;                     // MAX over ranks, as counted by RCCL
    const double total = (double)signals * (double)len;
    const double ms = dt / steps * 1e3;
    printf("{\"metric\": \"Msamples/s, batched column-wise db4 dwt %lld x %lld f32 sharded over the GPUs (BASELINE.json configs[4])\", "
           "\"value\": %.1f, \"unit\": \"Msamples/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.5f, "
           "\"higher_is_better\": true, \"scaling\": \"strong\", \"vs_baseline\": null, \"dtype\": \"f32\", "
           "\"data\": \"synthetic (splitmix64, generated on each device, seed 4242 + 1000*rank), resident in HBM\", "
           "\"config\": {\"workload\": \"batched column-wise dwt %s %lld x %lld f32, L=%d, %d shards\", \"signals_total\": %lld, "
           "\"signals_per_rank\": %lld, \"signals_covered_all_ranks\": %lld, "
           "\"parallelism\": \"column block partition over %d devices, ONE process, one host thread + one wl_ctx per device, no data-path collective\", "
           "\"collectives\": \"ncclCommInitAll; 1 ncclBroadcast of the wavelet (2048 B); ncclAllReduce SUM (checksum) and MAX (time) of 8 B\", "
           "\"kernel\": \"%s\", \"taps_received_by_last_rank\": %d}, "
           "\"achieved_hbm_GBps_algorithmic\": %.1f, \"checksum_all_ranks\": %.9g, \"host\": \"native (tools/wlbench_mgpu.cpp, no torch)\", "
           "\"rccl\": {\"version_code\": %d, \"ranks\": %d}}\n",
           signals, len, total / (dt / steps) / 1e6, world, steps, warmup, ms, kv["filt"].c_str(), signals, len, L, world, signals,
           res[0].hi - res[0].lo, covered, world, res[0].kernel.c_str(), res[world - 1].taps_received,
           8.0 * total / (dt / steps) / 1e9, res[0].checksum_all, rccl_version, world);
    return 0;
}
