// dense_fused_bench.cpp -- first contact + timing of the package's dense kernels without Python (a cold `import torch` costs a
// minute of a GPU call).  r06: binds the libraries with dlopen, so that the r05 kernels (tools/micro/r05/, built as their own
// library) run beside the current ones on the same data:
//   forward  Y = relu(X.W^T) (+ sign mask), input gradient (G (.) mask).W (+ Gm), weight gradient Gm^T.X (bf16 planes and, when
//   the probe entry point was built, the fp32-MFMA form)
// at the benchmark layer shape (n = 232 965, f = 128), the papers width (f = 64) and ragged shapes; checks sampled rows / entries
// against float64 on the host (bound: 2e-6 of sum |a||b|, the mask exactly) and times K launches of each with HIP events.
// One JSON line per case; exit code 1 on any mismatch.
//   tools/micro/build_dense_fused_bench.sh && tools/micro/dense_fused_bench.bin [n] [reps]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static float rnd() {                      // xorshift, uniform in (-1, 1)
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((double)(rng_state >> 11) / (double)(1ull << 53) * 2.0 - 1.0);
}

typedef int (*fwd_fn)(const float *, int64_t, int64_t, int32_t, const float *, int64_t, int32_t, float *, int64_t, int32_t, uint32_t *, void *);
typedef int (*bwd_fn)(const float *, int64_t, const uint32_t *, float *, int64_t, int64_t, int32_t, const float *, int64_t, int32_t, float *,
                      int64_t, void *);
typedef int (*wg_fn)(const float *, int64_t, const float *, int64_t, int64_t, int32_t, int32_t, float *, int64_t, float *, int64_t, void *);
typedef int64_t (*wgws_fn)(void);
typedef const char *(*err_fn)(void);
// the r05 library's signatures (no mask: the backward reads Y)
typedef int (*fwd5_fn)(const float *, int64_t, int64_t, int32_t, const float *, int64_t, int32_t, float *, int64_t, int32_t, void *);
typedef int (*bwd5_fn)(const float *, int64_t, const float *, int64_t, float *, int64_t, int64_t, int32_t, const float *, int64_t, int32_t, float *,
                       int64_t, void *);

struct Lib {
    fwd_fn fwd = nullptr; bwd_fn bwd = nullptr; wg_fn wg = nullptr, wg32 = nullptr; wgws_fn wgws = nullptr; err_fn err = nullptr, werr = nullptr;
    fwd5_fn fwd5 = nullptr; bwd5_fn bwd5 = nullptr;
};

// PGCN_BENCH_HEATER=<MB>: a device-to-device copy of that many MB runs on the stream before EVERY timed launch and only the launch
// itself is bracketed by events -- the state a dense kernel meets inside an epoch, right behind 1.5 ms of memory-bound SpMM kernels
// (clocks chosen for those, caches full of something else), instead of 20 launches back to back on an idle chip.
static char *g_heat_src = nullptr, *g_heat_dst = nullptr;
static size_t g_heat_bytes = 0;

template <class F>
static double time_us(hipStream_t s, int reps, F &&f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) f();
    double us = 0;
    if (g_heat_bytes) {
        for (int r = 0; r < reps; ++r) {
            CK(hipMemcpyAsync(g_heat_dst, g_heat_src, g_heat_bytes, hipMemcpyDeviceToDevice, s));
            CK(hipEventRecord(a, s));
            f();
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            us += ms * 1e3;
        }
        us /= reps;
    } else {
        CK(hipEventRecord(a, s));
        for (int r = 0; r < reps; ++r) f();
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        us = ms * 1e3 / reps;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return us;
}

static int run_case(const Lib &L, int64_t n, int fin, int fout, int reps, bool time_it) {
    std::vector<float> X((size_t)n * fin), W((size_t)fout * fin), G((size_t)n * fout);
    for (auto &v : X) v = rnd();
    for (auto &v : W) v = rnd() * 0.1f;
    for (auto &v : G) v = rnd();
    const int mw = (fout + 31) / 32;
    float *dX_, *dW_, *dY_, *dG_, *dGm_, *dDX_, *dDW_, *dWS_;
    uint32_t *dM_;
    const int64_t ws_elems = L.wgws ? L.wgws() : 0;
    CK(hipMalloc(&dX_, X.size() * 4 + 16)); CK(hipMalloc(&dW_, W.size() * 4)); CK(hipMalloc(&dY_, (size_t)n * fout * 4 + 16));
    CK(hipMalloc(&dG_, G.size() * 4 + 16)); CK(hipMalloc(&dGm_, G.size() * 4 + 16)); CK(hipMalloc(&dDX_, X.size() * 4 + 16));
    CK(hipMalloc(&dM_, (size_t)n * mw * 4 + 16)); CK(hipMalloc(&dDW_, W.size() * 4)); CK(hipMalloc(&dWS_, (size_t)(ws_elems ? ws_elems : 4) * 4));
    CK(hipMemcpy(dX_, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW_, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dG_, G.data(), G.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dY_, 0xff, (size_t)n * fout * 4)); CK(hipMemset(dGm_, 0xff, G.size() * 4)); CK(hipMemset(dDX_, 0xff, X.size() * 4));
    CK(hipMemset(dM_, 0x55, (size_t)n * mw * 4)); CK(hipMemset(dDW_, 0xff, W.size() * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int rc = L.fwd(dX_, fin, n, fin, dW_, fin, fout, dY_, fout, 1, dM_, s);
    if (rc) { fprintf(stderr, "forward rc %d: %s\n", rc, L.err()); return 1; }
    rc = L.bwd(dG_, fout, dM_, dGm_, fout, n, fout, dW_, fin, fin, dDX_, fin, s);
    if (rc) { fprintf(stderr, "backward rc %d: %s\n", rc, L.err()); return 1; }
    if (L.wg) {
        rc = L.wg(dGm_, fout, dX_, fin, n, fout, fin, dDW_, fin, dWS_, ws_elems, s);
        if (rc) { fprintf(stderr, "weight gradient rc %d: %s\n", rc, L.werr()); return 1; }
    }
    CK(hipStreamSynchronize(s));
    std::vector<float> Y((size_t)n * fout), Gm(G.size()), DX(X.size()), DW(W.size());
    std::vector<uint32_t> M((size_t)n * mw);
    CK(hipMemcpy(Y.data(), dY_, Y.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Gm.data(), dGm_, Gm.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(DX.data(), dDX_, DX.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(M.data(), dM_, M.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(DW.data(), dDW_, DW.size() * 4, hipMemcpyDeviceToHost));
    // rows to check: the first and last 64 (tile edges, the ragged tail) and a stride through the middle
    double ef = 0, eb = 0, ew = 0;
    int64_t bad_mask = 0, bad_bits = 0, rows = 0;
    const int64_t step = n > 4096 ? n / 1500 : 1;
    for (int64_t i = 0; i < n; i += (i < 64 || i >= n - 65 ? 1 : step)) {
        ++rows;
        for (int o = 0; o < fout; ++o) {
            double sum = 0, den = 0;
            for (int k = 0; k < fin; ++k) { sum += (double)X[i * fin + k] * W[(size_t)o * fin + k]; den += fabs((double)X[i * fin + k] * W[(size_t)o * fin + k]); }
            const double want = sum > 0 ? sum : 0;
            const double e = fabs((double)Y[i * fout + o] - want) / (den + 1e-30);
            if (!(e <= ef)) ef = e;                                   // (NaN-catching comparison)
            const float gm = Y[i * fout + o] > 0.f ? G[i * fout + o] : 0.f;
            if (!(Gm[i * fout + o] == gm)) ++bad_mask;
            const bool bit = (M[i * mw + o / 32] >> (o % 32)) & 1u;
            if (bit != (Y[i * fout + o] > 0.f)) ++bad_bits;
        }
        for (int o = fout; o < 32 * mw; ++o) if ((M[i * mw + o / 32] >> (o % 32)) & 1u) ++bad_bits;       // bits beyond the width: zero
        for (int k = 0; k < fin; ++k) {
            double sum = 0, den = 0;
            for (int o = 0; o < fout; ++o) { const double gm = Y[i * fout + o] > 0.f ? G[i * fout + o] : 0.f; sum += gm * W[(size_t)o * fin + k]; den += fabs(gm * W[(size_t)o * fin + k]); }
            const double e = fabs((double)DX[i * fin + k] - sum) / (den + 1e-30);
            if (!(e <= eb)) eb = e;
        }
    }
    if (L.wg) {                                  // entries of dW: a stride through the matrix, all rows in float64
        const int es = fout * fin > 400 ? fout * fin / 199 : 1;
        for (int e0 = 0; e0 < fout * fin; e0 += es) {
            const int o = e0 / fin, k = e0 % fin;
            double sum = 0, den = 0;
            for (int64_t i = 0; i < n; ++i) { const double p = (double)Gm[i * fout + o] * X[i * fin + k]; sum += p; den += fabs(p); }
            const double e = fabs((double)DW[(size_t)o * fin + k] - sum) / (den + 1e-30);
            if (!(e <= ew)) ew = e;
        }
    }
    double us_f = 0, us_f0 = 0, us_b = 0, us_b0 = 0, us_w = 0, us_w32 = 0, us_f5 = 0, us_b5 = 0;
    if (time_it) {
        // three interleaved rounds of every kernel, the median of each (the clock sags as the chip warms up: what runs first looks best)
        std::vector<double> t[8];
        for (int round = 0; round < 3; ++round) {
            if (L.fwd5) t[6].push_back(time_us(s, reps, [&] { L.fwd5(dX_, fin, n, fin, dW_, fin, fout, dY_, fout, 1, s); }));
            t[0].push_back(time_us(s, reps, [&] { L.fwd(dX_, fin, n, fin, dW_, fin, fout, dY_, fout, 1, dM_, s); }));
            t[1].push_back(time_us(s, reps, [&] { L.fwd(dX_, fin, n, fin, dW_, fin, fout, dY_, fout, 1, nullptr, s); }));
            if (L.bwd5) t[7].push_back(time_us(s, reps, [&] { L.bwd5(dG_, fout, dY_, fout, dGm_, fout, n, fout, dW_, fin, fin, dDX_, fin, s); }));
            t[2].push_back(time_us(s, reps, [&] { L.bwd(dG_, fout, dM_, dGm_, fout, n, fout, dW_, fin, fin, dDX_, fin, s); }));
            t[3].push_back(time_us(s, reps, [&] { L.bwd(dG_, fout, dM_, nullptr, 0, n, fout, dW_, fin, fin, dDX_, fin, s); }));
            if (L.wg) t[4].push_back(time_us(s, reps, [&] { L.wg(dGm_, fout, dX_, fin, n, fout, fin, dDW_, fin, dWS_, ws_elems, s); }));
            if (L.wg32) t[5].push_back(time_us(s, reps, [&] { L.wg32(dGm_, fout, dX_, fin, n, fout, fin, dDW_, fin, dWS_, ws_elems, s); }));
        }
        auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        us_f = med(t[0]); us_f0 = med(t[1]); us_b = med(t[2]); us_b0 = med(t[3]); us_w = med(t[4]); us_w32 = med(t[5]); us_f5 = med(t[6]); us_b5 = med(t[7]);
    }
    const bool ok = ef <= 2e-6 && eb <= 2e-6 && ew <= 2e-6 && bad_mask == 0 && bad_bits == 0;
    const double bytes_f = (double)n * (fin + fout) * 4, bytes_b = (double)n * (2.0 * fout + fin) * 4 + (double)n * mw * 4,
                 bytes_w = (double)n * (fin + fout) * 4;
    printf("{\"n\": %lld, \"fin\": %d, \"fout\": %d, \"rows_checked\": %lld, \"err_forward\": %.3g, \"err_input_grad\": %.3g, \"err_weight_grad\": %.3g, "
           "\"mask_mismatches\": %lld, \"mask_bit_mismatches\": %lld, \"ok\": %s, ",
           (long long)n, fin, fout, (long long)rows, ef, eb, ew, (long long)bad_mask, (long long)bad_bits, ok ? "true" : "false");
    printf("\"forward_us\": %.1f, \"forward_GBps\": %.0f, \"forward_no_mask_us\": %.1f, \"input_grad_us\": %.1f, \"input_grad_GBps\": %.0f, "
           "\"input_grad_no_gm_us\": %.1f, \"weight_grad_us\": %.1f, \"weight_grad_GBps\": %.0f, \"weight_grad_f32mfma_us\": %.1f, "
           "\"r05_forward_us\": %.1f, \"r05_input_grad_us\": %.1f}\n",
           us_f, us_f > 0 ? bytes_f / us_f / 1e3 : 0.0, us_f0, us_b, us_b > 0 ? bytes_b / us_b / 1e3 : 0.0, us_b0, us_w,
           us_w > 0 ? bytes_w / us_w / 1e3 : 0.0, us_w32, us_f5, us_b5);
    fflush(stdout);
    hipFree(dX_); hipFree(dW_); hipFree(dY_); hipFree(dG_); hipFree(dGm_); hipFree(dDX_); hipFree(dM_); hipFree(dDW_); hipFree(dWS_);
    hipStreamDestroy(s);
    return ok ? 0 : 1;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 232965;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    std::string here = argv[0];
    here = here.substr(0, here.find_last_of('/') + 1);
    const std::string cur = argc > 3 ? argv[3] : here + "../../scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd/lib/libpgcn_gemm.so";
    const std::string old = here + "libpgcn_dense_r05.so", probe = here + "libpgcn_wgrad_probe.so";
    void *h = dlopen(cur.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", cur.c_str(), dlerror()); return 2; }
    Lib L;
    L.fwd = (fwd_fn)dlsym(h, "pgcn_linear_relu_f32"); L.bwd = (bwd_fn)dlsym(h, "pgcn_linear_relu_grad_input_f32");
    L.wg = (wg_fn)dlsym(h, "pgcn_linear_weight_grad_f32"); L.wgws = (wgws_fn)dlsym(h, "pgcn_linear_weight_grad_ws_elems");
    L.err = (err_fn)dlsym(h, "pgcn_dense_last_error"); L.werr = (err_fn)dlsym(h, "pgcn_wgrad_last_error");
    if (!L.fwd || !L.bwd || !L.err) { fprintf(stderr, "entry points missing in %s\n", cur.c_str()); return 2; }
    if (void *hp = dlopen(probe.c_str(), RTLD_NOW | RTLD_LOCAL)) L.wg32 = (wg_fn)dlsym(hp, "pgcn_linear_weight_grad_f32mfma_f32");
    if (void *ho = dlopen(old.c_str(), RTLD_NOW | RTLD_LOCAL)) {
        L.fwd5 = (fwd5_fn)dlsym(ho, "pgcn_linear_relu_f32"); L.bwd5 = (bwd5_fn)dlsym(ho, "pgcn_linear_relu_grad_input_f32");
    }
    if (const char *h = getenv("PGCN_BENCH_HEATER")) {
        g_heat_bytes = (size_t)atoll(h) << 20;
        if (g_heat_bytes) { CK(hipMalloc(&g_heat_src, g_heat_bytes)); CK(hipMalloc(&g_heat_dst, g_heat_bytes)); CK(hipMemset(g_heat_src, 1, g_heat_bytes)); }
        printf("# heater: %zu MB copied device to device before every timed launch\n", g_heat_bytes >> 20);
    }
    int fails = 0;
    rng_state = 0x9e3779b97f4a7c15ull;
    fails += run_case(L, n, 128, 128, reps, true);            // the benchmark layer
    Lib Lc = L; Lc.fwd5 = nullptr; Lc.bwd5 = nullptr;
    fails += run_case(Lc, 1000, 128, 40, 1, false);           // ragged widths / row counts
    fails += run_case(Lc, 77, 64, 128, 1, false);
    fails += run_case(Lc, 4097, 36, 100, 1, false);
    fails += run_case(Lc, 33, 4, 4, 1, false);
    fails += run_case(Lc, 4096, 128, 128, 1, false);          // no ragged tile at all
    fails += run_case(Lc, 15, 128, 64, 1, false);             // less than one step of the weight gradient
    rng_state = 0x9e3779b97f4a7c15ull;
    fails += run_case(L, n, 64, 64, reps, true);              // the papers shape's width
    return fails ? 1 : 0;
}
