// wlbench -- torch-free timing harness over the C ABI (profiling runs on the GPU box start in
// milliseconds instead of waiting for `import torch`).  Not product code.
//
//   wlbench [key=value ...]
//     n0=8192 n1=8192 n2=1   array extents (Julia order, n0 fastest); nd = number of extents > 1 ... or nd=
//     L=13                   levels (0 => maxtransformlevels over the transformed dims)
//     filt=db4|db2|haar|sym5|db6   orthogonal filter (taps below are wt.py's values)
//     dtype=f32|f64
//     fw=1                   1 forward, 0 inverse
//     reps=50 warm=20
//     mode=seq|each          seq: one event pair around all reps;  each: one pair per call (avg/med/min)
//     check=1                print a checksum of y
//     opt=KEY:VAL,...        wl_ctx_set_option pairs
//     dwtc=1                 batched column-wise transform of n1 signals of length n0 (wl_dwtc_filter)
//     rot=1                  number of distinct input arrays the calls rotate over (bench.py rotates 3 x 256 MiB for C3)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "../include/wavelets_mi355x.h"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "hip error %d (%s) at line %d\n", (int)e__, hipGetErrorString(e__), __LINE__); return 2; } } while (0)

static const std::map<std::string, std::vector<double>> kTaps = {
    {"haar", {0.7071067811865476, 0.7071067811865476}},
    {"db2", {0.4829629131445342, 0.8365163037378079, 0.2241438680420133, -0.12940952255126045}},
    {"db3", {0.3326705529500827, 0.8068915093110927, 0.45987750211849154, -0.13501102001025467, -0.08544127388202664, 0.03522629188570957}},
    {"db4", {0.23037781330889648, 0.7148465705529157, 0.6308807679298589, -0.027983769416860003, -0.18703481171909309, 0.030841381835560722, 0.03288301166688518, -0.010597401785069035}},
    {"sym5", {0.019538882735386898, -0.02110183402492983, -0.17532808990810747, 0.016602105764424325, 0.633978963456949, 0.7234076904038076, 0.1993975339769955, -0.03913424930258344, 0.0295194909260734, 0.02733306834516448}},
    {"db6", {0.11154074335010944, 0.4946238903984531, 0.7511339080210953, 0.31525035170919813, -0.22626469396543938, -0.12976686756726197, 0.09750160558732307, 0.027522865530305647, -0.031582039317485995, 0.0005538422011614999, 0.004777257510945508, -0.0010773010853084798}},
    {"db7", {0.07785205408500918, 0.3965393194819173, 0.7291320908462351, 0.46978228740519296, -0.14390600392856484, -0.2240361849938754, 0.07130921926683056, 0.08061260915108302, -0.0380299369350144, -0.016574541630666913, 0.012550998556099856, 0.00042957797292136554, -0.001801640704047492, 0.00035371379997452024}},
    {"db9", {0.038077947363878366, 0.24383467461259042, 0.6048231236901115, 0.6572880780513005, 0.13319738582500773, -0.2932737832791742, -0.09684078322297636, 0.14854074933810593, 0.030725681479334035, -0.06763282906133081, 0.00025094711483188403, 0.02236166212367899, -0.0047232047577513816, -0.004281503682463445, 0.0018476468830562337, 0.00023038576352319562, -0.0002519631889427106, 3.934732031627169e-05}},
    {"db10", {0.026670057900555565, 0.1881768000776916, 0.5272011889317259, 0.6884590394536039, 0.2811723436605773, -0.24984642432731435, -0.195946274377378, 0.12736934033579372, 0.09305736460357084, -0.07139414716639567, -0.029457536821876806, 0.03321267405934165, 0.0036065535669558176, -0.010733175483330465, 0.001395351747052877, 0.0019924052951850644, -0.0006858566949597138, -0.00011646685512928556, 9.358867032006975e-05, -1.3264202894521261e-05}},
    {"sym8", {-0.0033824159513594415, -0.0005421323316355467, 0.031695087810345246, 0.0076074873252847675, -0.14329423835105426, -0.06127335906790878, 0.48135965125920116, 0.7771857516997479, 0.3644418948359564, -0.0519458381078751, -0.02721902991681368, 0.049137179673476784, 0.003808752014060054, -0.014952258336792626, -0.00030292051455164, 0.0018899503329007496}},
    {"coif4", {0.0163873364635998, -0.0414649367819558, -0.0673725547222826, 0.3861100668229939, 0.8127236354493977, 0.4170051844236707, -0.0764885990786692, -0.0594344186467388, 0.0236801719464464, 0.0056114348194211, -0.0018232088707116, -0.0007205494453679}},
    {"coif6", {-0.003793512864491, 0.0077825964273254, 0.0234526961418362, -0.0657719112818552, -0.0611233900026726, 0.405176902409615, 0.7937772226256169, 0.4284834763776168, -0.0717998216193117, -0.0823019271068856, 0.0345550275730615, 0.0158805448636158, -0.0090079761366615, -0.0025745176887502, 0.0011175187708906, 0.0004662169601129, -7.09833031381e-05, -3.45997728362e-05}},
    {"beyl", {0.09930576537400788, 0.4242153608130337, 0.6998252140570556, 0.4497182511490357, -0.11092759834800882, -0.264497231446021, 0.026900308804002133, 0.15553873187701237, -0.01752074626700139, -0.08854363062300702, 0.01967986604400156, 0.042916387274003404, -0.017460408696001385, -0.01436580796900114, 0.010040411845000796, 0.0014842347820001177, -0.0027360316260002173, 0.0006404853290000508}},
    {"db8", {0.05441584224310398, 0.3128715909142999, 0.6756307362972896, 0.5853546836542072, -0.015829105256348633, -0.2840155429615473, 0.00047248457391376, 0.1287474266204779, -0.017369301001807197, -0.04408825393079476, 0.013981027917398216, 0.008746094047405775, -0.004870352993451561, -0.0003917403733769489, 0.0006754494064505684, -0.00011747678412476926}},
};

__global__ void k_fill(float *p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (float)((double)(z >> 11) * (1.0 / 9007199254740992.0)) - 0.5f;
    }
}
__global__ void k_fill64(double *p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}

int main(int argc, char **argv)
{
    std::map<std::string, std::string> kv = {{"n0", "8192"}, {"n1", "8192"}, {"n2", "1"}, {"L", "0"}, {"filt", "db4"}, {"dtype", "f32"},
                                             {"fw", "1"}, {"reps", "50"}, {"warm", "20"}, {"mode", "seq"}, {"check", "1"}, {"opt", ""},
                                             {"path", "0"}, {"rot", "1"}};
    for (int i = 1; i < argc; ++i) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { fprintf(stderr, "bad arg %s\n", argv[i]); return 1; }
        kv[std::string(argv[i], eq - argv[i])] = eq + 1;
    }
    int64_t dims[3] = {atoll(kv["n0"].c_str()), atoll(kv["n1"].c_str()), atoll(kv["n2"].c_str())};
    int nd = dims[2] > 1 ? 3 : (dims[1] > 1 ? 2 : 1);
    if (kv.count("nd")) nd = atoi(kv["nd"].c_str());
    const int dtype = kv["dtype"] == "f64" ? WL_F64 : WL_F32;
    const size_t es = dtype == WL_F64 ? 8 : 4;
    int L = atoi(kv["L"].c_str());
    if (L == 0) {
        L = 99;
        for (int d = 0; d < nd; ++d) L = std::min(L, wl_maxtransformlevels(dims[d]));
    }
    const int fw = atoi(kv["fw"].c_str()), reps = atoi(kv["reps"].c_str()), warm = atoi(kv["warm"].c_str());
    // filt=cdf97lift: the cdf9/7 lifting scheme (wt_main.jl:451-480 table order) through wl_dwt_lifting_oop
    const bool lifting = kv["filt"] == "cdf97lift";
    static const int32_t ls_upd[4] = {1, 0, 1, 0}, ls_nc[4] = {2, 2, 2, 2}, ls_sh[4] = {0, 1, 0, 1};
    static const double ls_c[8] = {1.5861343420604, 1.5861343420604, 0.05298011857291494, 0.05298011857291494,
                                   -0.882911075531393, -0.882911075531393, -0.44350685204384654, -0.44350685204384654};
    auto it = kTaps.find(lifting ? std::string("db4") : kv["filt"]);
    if (it == kTaps.end()) { fprintf(stderr, "unknown filter\n"); return 1; }
    const std::vector<double> &qmf = it->second;
    const size_t N = (size_t)dims[0] * dims[1] * dims[2];

    wl_ctx *ctx = nullptr;
    int rc = wl_ctx_create(0, &ctx);
    if (rc) { fprintf(stderr, "wl_ctx_create: %s\n", wl_strerror(rc)); return 2; }
    wl_ctx_set_path(ctx, atoi(kv["path"].c_str()));
#ifdef WL_HAVE_OPTIONS
    {
        std::string o = kv["opt"];
        size_t p = 0;
        while (p < o.size()) {
            size_t c = o.find(',', p);
            if (c == std::string::npos) c = o.size();
            std::string item = o.substr(p, c - p);
            size_t col = item.find(':');
            if (col != std::string::npos) {
                int r = wl_ctx_set_option(ctx, item.substr(0, col).c_str(), atoll(item.substr(col + 1).c_str()));
                if (r) fprintf(stderr, "option %s: %s\n", item.c_str(), wl_strerror(r));
            }
            p = c + 1;
        }
    }
#endif
    void *x = nullptr, *y = nullptr;
    const int rot = std::max(1, atoi(kv["rot"].c_str()));
    std::vector<void *> xs(rot, nullptr);
    for (int r = 0; r < rot; ++r) {
        CK(hipMalloc(&xs[r], N * es));
        if (dtype == WL_F32) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (float *)xs[r], N, 42u + 977u * r);
        else hipLaunchKernelGGL(k_fill64, dim3(4096), dim3(256), 0, 0, (double *)xs[r], N, 42u + 977u * r);
    }
    x = xs[0];
    CK(hipMalloc(&y, N * es));
    CK(hipMemset(y, 0, N * es));
    rc = wl_ctx_reserve(ctx, lifting ? wl_workspace_bytes_full(dtype, nd, dims, L) : wl_workspace_bytes(dtype, nd, dims, L));
    if (rc) { fprintf(stderr, "reserve: %s\n", wl_strerror(rc)); return 2; }
    CK(hipDeviceSynchronize());
    const bool batched = kv.count("dwtc") && atoi(kv["dwtc"].c_str()) != 0;      // dwtc=1: n1 signals of length n0 (columns)
    if (batched && atoi(kv["L"].c_str()) == 0) L = wl_maxtransformlevels(dims[0]);
    int ncall = 0;
    auto call = [&]() {
        x = xs[ncall++ % rot];
        if (lifting) return wl_dwt_lifting_oop(ctx, dtype, y, x, nd, dims, 4, ls_upd, ls_nc, ls_sh, ls_c, 1.1496043988603355, 0.8698644516247099, L, fw, nullptr);
        if (batched) return wl_dwtc_filter(ctx, dtype, y, x, dims[0], dims[1], dims[0], qmf.data(), (int)qmf.size(), L, fw, nullptr);
        return wl_dwt_filter(ctx, dtype, y, x, nd, dims, qmf.data(), (int)qmf.size(), L, fw, nullptr);
    };
    for (int i = 0; i < warm; ++i) {
        rc = call();
        if (rc) { fprintf(stderr, "wl_dwt_filter: %s (hip %d)\n", wl_strerror(rc), wl_last_hip_error(ctx)); return 2; }
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double avg = 0, med = 0, mn = 0;
    if (kv["mode"] == "each") {
        std::vector<float> v;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0, nullptr));
            call();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.push_back(ms);
        }
        std::sort(v.begin(), v.end());
        for (float f : v) avg += f;
        avg /= v.size(); med = v[v.size() / 2]; mn = v[0];
    } else {
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) call();
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        avg = med = mn = ms / reps;
    }
    double sum = 0;
    if (atoi(kv["check"].c_str())) {
        std::vector<char> h(std::min<size_t>(N, (size_t)1 << 22) * es);
        CK(hipMemcpy(h.data(), y, h.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size() / es; ++i) sum += dtype == WL_F64 ? ((double *)h.data())[i] : (double)((float *)h.data())[i];
    }
    const double alg = 2.0 * N * es;
    printf("{\"n\": [%lld, %lld, %lld], \"nd\": %d, \"L\": %d, \"filt\": \"%s\", \"dtype\": \"%s\", \"fw\": %d, \"mode\": \"%s\", \"reps\": %d, "
           "\"avg_us\": %.2f, \"med_us\": %.2f, \"min_us\": %.2f, \"alg_TBps\": %.3f, \"frac8\": %.4f, \"kernel\": \"%s\", \"opt\": \"%s\", \"sum\": %.9g}\n",
           (long long)dims[0], (long long)dims[1], (long long)dims[2], nd, L, kv["filt"].c_str(), kv["dtype"].c_str(), fw, kv["mode"].c_str(), reps,
           avg * 1e3, med * 1e3, mn * 1e3, alg / (avg * 1e-3) / 1e12, alg / (avg * 1e-3) / 8e12, wl_last_kernel(ctx), kv["opt"].c_str(), sum);
    // wgtime=DIR: experiment builds with -DWL_WGTIME export per-workgroup stamps of the last launch of each instrumented kernel family
    // (8 words per workgroup) -> DIR/<family>.txt
    if (kv.count("wgtime")) {
        typedef int (*fn_t)(unsigned long long *, size_t);
        for (const char *fam : {"pair", "tileB", "tile", "tail", "ftile"}) {
            const std::string sym = std::string("wl_debug_wgtimes_") + fam;
            fn_t fn = (fn_t)dlsym(RTLD_DEFAULT, sym.c_str());
            if (!fn) continue;
            std::vector<unsigned long long> t(8 * 8192);
            if (fn(t.data(), t.size())) { fprintf(stderr, "%s failed\n", sym.c_str()); return 3; }
            FILE *f = fopen((kv["wgtime"] + "/" + fam + ".txt").c_str(), "w");
            if (!f) { fprintf(stderr, "cannot write to %s\n", kv["wgtime"].c_str()); return 3; }
            for (size_t i = 0; i < 8192; ++i) {
                if (!t[8 * i]) continue;
                fprintf(f, "%zu", i);
                for (int k = 0; k < 8; ++k) fprintf(f, " %llu", t[8 * i + k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    wl_ctx_destroy(ctx);
    return 0;
}
