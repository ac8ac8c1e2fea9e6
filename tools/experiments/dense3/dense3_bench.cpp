// Stand-alone harness (r03) of pgcn_spmm_dense_bf16x3_f32 next to pgcn_spmm_dense_f32: the same random 128 x 128
// tiles (fp32 swizzle for the fp32-MFMA kernel, three bf16 planes for the bf16 one), a Reddit-sized operand
// (232 965 x 128), pieces of 3 / 6 / 12 tiles.  Prints the time per launch and per tile and CU of both kernels,
// the error of both against a float64 loop relative to sum |a||h| per output, run-to-run determinism, and three
// edge cases of the new kernel (f = 72 with an odd leading dimension, the last partial panel, an Inf in an
// operand row that only structural zeros touch).  A pure HIP binary: no torch, a gpurun call costs ~20 s.
//   tools/experiments/dense3/build.sh   (compiles the kernel three times -- real, two probes -- and links lib/libpgcn_hip.so for the
//   fp32 kernel and the error helpers; the binary travels to the GPU box with the snapshot)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

extern "C" int pgcn_spmm_dense_f32(const int32_t *, int64_t, const int32_t *, const float *, const float *, int64_t, int64_t,
                                   int32_t, float *, int64_t, int64_t, void *);
extern "C" int pgcn_spmm_dense_bf16x3_f32(const int32_t *, int64_t, const int32_t *, const void *, const float *, int64_t,
                                          int64_t, int32_t, float *, int64_t, int64_t, void *);
// the kernel source is compiled into this binary: as it is, and with PGCN_DENSE3_PROBE = 1 / 2 (timing only, see the kernel file)
extern "C" int pgcn_spmm_dense_bf16x3_probe1_f32(const int32_t *, int64_t, const int32_t *, const void *, const float *, int64_t,
                                                 int64_t, int32_t, float *, int64_t, int64_t, void *);
extern "C" int pgcn_spmm_dense_bf16x3_probe2_f32(const int32_t *, int64_t, const int32_t *, const void *, const float *, int64_t,
                                                 int64_t, int32_t, float *, int64_t, int64_t, void *);
// ... and with PGCN_DENSE3_PROBE = 3: the real kernel with per-wave phase timers
extern "C" int pgcn_spmm_dense_bf16x3_probe3_f32(const int32_t *, int64_t, const int32_t *, const void *, const float *, int64_t,
                                                 int64_t, int32_t, float *, int64_t, int64_t, void *);
extern "C" int pgcn_dense3_set_timers(void *);
extern "C" const char *pgcn_last_error(void);

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define PCHECK(x) do { int r_ = (x); if (r_ != 0) { printf("pgcn error %d (%s) at line %d\n", r_, pgcn_last_error(), __LINE__); exit(1); } } while (0)

static uint16_t bf16_rne(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; }

struct Case {
    int64_t ncols; int f; int64_t ldb; int ntiles;
    std::vector<float> tiles;          // [ntiles][128][128] row-major (i, k)
    std::vector<int32_t> panel;        // [ntiles]
    std::vector<float> B;              // [ncols][ldb]
    std::vector<float> vals;           // fp32 swizzle
    std::vector<uint16_t> planes;      // bf16 planes
};

static void layouts(Case &c) {
    c.vals.assign((size_t)c.ntiles * 16384, 0.f);
    c.planes.assign((size_t)c.ntiles * 16384 * 3, 0);
    for (int t = 0; t < c.ntiles; ++t)
        for (int i = 0; i < 128; ++i)
            for (int k = 0; k < 128; ++k) {
                const float a = c.tiles[((size_t)t * 128 + i) * 128 + k];
                const int w = i / 32, il = i % 32, s = k / 2, kh = k % 2;
                c.vals[(size_t)t * 16384 + ((w * 16 + s / 4) * 64 + kh * 32 + il) * 4 + s % 4] = a;
                const uint16_t a1 = bf16_rne(a);
                const float r1 = a - bf16_f32(a1);
                const uint16_t a2 = bf16_rne(r1);
                const uint16_t a3 = bf16_rne(r1 - bf16_f32(a2));
                const int ks = k / 16, hk = (k / 8) % 2, j = k % 8;
                const size_t e = (((((size_t)t * 4 + w) * 8 + ks) * 3) * 64 + hk * 32 + il) * 8 + j;
                c.planes[e] = a1; c.planes[e + 512] = a2; c.planes[e + 1024] = a3;
            }
}

static void make_case(Case &c, int64_t ncols, int f, int64_t ldb, int ntiles, double fill, unsigned seed, bool last_panel_first) {
    c.ncols = ncols; c.f = f; c.ldb = ldb; c.ntiles = ntiles;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const int npanels = (int)((ncols + 127) / 128);
    c.panel.resize(ntiles);
    c.tiles.assign((size_t)ntiles * 16384, 0.f);
    for (int t = 0; t < ntiles; ++t) {
        c.panel[t] = (int32_t)(rng() % npanels);
        if (last_panel_first && t == 1) c.panel[t] = npanels - 1;           // the partial panel: rows >= ncols read as zero
        const int64_t valid = std::min<int64_t>(128, ncols - (int64_t)c.panel[t] * 128);
        for (int i = 0; i < 128; ++i)
            for (int k = 0; k < valid; ++k)
                if (U(rng) < fill) c.tiles[((size_t)t * 128 + i) * 128 + k] = (0.01f + U(rng)) * 0.05f * (rng() & 1 ? 1.f : -1.f);
    }
    c.B.resize((size_t)ncols * ldb);
    for (auto &x : c.B) x = 2.f * U(rng) - 1.f;
    layouts(c);
}

struct Dev { int32_t *work, *panel; float *vals, *B, *ws; void *planes; };

static double ref_err(const Case &c, const std::vector<int32_t> &work, const std::vector<float> &got, const std::vector<int> &pieces) {
    double worst = 0;
    for (int p : pieces) {
        const int t0 = work[4 * p + 1], nt = work[4 * p + 2], slot = work[4 * p + 3];
        for (int i = 0; i < 128; ++i)
            for (int n = 0; n < c.f; ++n) {
                double s = 0, sa = 0;
                for (int t = t0; t < t0 + nt; ++t)
                    for (int k = 0; k < 128; ++k) {
                        const float a = c.tiles[((size_t)t * 128 + i) * 128 + k];
                        if (a == 0.f) continue;
                        const double h = c.B[((size_t)c.panel[t] * 128 + k) * c.ldb + n];
                        s += (double)a * h; sa += std::fabs((double)a * h);
                    }
                const double g = got[((size_t)slot + i) * c.f + n];
                const double e = std::isfinite(g) ? std::fabs(g - s) / (sa > 0 ? sa : 1.0) : 1e30;
                if (e > worst) worst = e;
            }
    }
    return worst;
}

static std::vector<int32_t> make_work(int ntiles, int per) {
    std::vector<int32_t> w;
    int p = 0;
    for (int t = 0; t < ntiles; t += per, ++p) { w.push_back(p % 1820); w.push_back(t); w.push_back(std::min(per, ntiles - t)); w.push_back(p * 128); }
    return w;
}

static void upload(const Case &c, Dev &d, int64_t max_slots) {
    CHECK(hipMalloc(&d.panel, c.panel.size() * 4)); CHECK(hipMemcpy(d.panel, c.panel.data(), c.panel.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d.vals, c.vals.size() * 4)); CHECK(hipMemcpy(d.vals, c.vals.data(), c.vals.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d.planes, c.planes.size() * 2)); CHECK(hipMemcpy(d.planes, c.planes.data(), c.planes.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d.B, c.B.size() * 4)); CHECK(hipMemcpy(d.B, c.B.data(), c.B.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d.ws, (size_t)max_slots * c.f * 4));
    CHECK(hipMalloc(&d.work, (size_t)c.ntiles * 16));
}
static void release(Dev &d) { hipFree(d.panel); hipFree(d.vals); hipFree(d.planes); hipFree(d.B); hipFree(d.ws); hipFree(d.work); }

static int launch(int which, const Case &c, const Dev &d, int64_t npieces) {
    const int64_t ns = npieces * 128;
    if (which == 0) return pgcn_spmm_dense_f32(d.work, npieces, d.panel, d.vals, d.B, c.ldb, c.ncols, c.f, d.ws, ns * c.f, ns, nullptr);
    if (which == 2) return pgcn_spmm_dense_bf16x3_probe1_f32(d.work, npieces, d.panel, d.planes, d.B, c.ldb, c.ncols, c.f, d.ws, ns * c.f, ns, nullptr);
    if (which == 5) return pgcn_spmm_dense_bf16x3_probe3_f32(d.work, npieces, d.panel, d.planes, d.B, c.ldb, c.ncols, c.f, d.ws, ns * c.f, ns, nullptr);
    if (which == 3) return pgcn_spmm_dense_bf16x3_probe2_f32(d.work, npieces, d.panel, d.planes, d.B, c.ldb, c.ncols, c.f, d.ws, ns * c.f, ns, nullptr);
    return pgcn_spmm_dense_bf16x3_f32(d.work, npieces, d.panel, d.planes, d.B, c.ldb, c.ncols, c.f, d.ws, ns * c.f, ns, nullptr);
}

int main(int argc, char **argv) {
    const int ntiles = argc > 1 ? atoi(argv[1]) : 3072;
    const double fill = argc > 2 ? atof(argv[2]) : 0.3;
    if (getenv("DENSE3_HOST_ONLY")) {      // (how long the host side takes, without a GPU)
        Case c; make_case(c, 232965, 128, 128, ntiles, fill, 7, true);
        Case c2 = c; layouts(c2);
        const std::vector<int32_t> work = make_work(ntiles, 6);
        std::vector<float> got((size_t)ntiles / 6 * 128 * 128 + 128 * 128, 0.f);
        printf("host only: err of an all-zero result %.3e\n", ref_err(c, work, got, {0, 1, 2, 500, 1000, 1001}));
        return 0;
    }
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    if (getenv("DENSE3_PMC")) {            // under rocprofv3 --pmc: two launches each of the fp32 and the bf16 kernel, 6 tiles per piece
        Case c; make_case(c, 232965, 128, 128, ntiles, fill, 7, true);
        Dev d; upload(c, d, (int64_t)ntiles * 128);
        const std::vector<int32_t> work = make_work(ntiles, 6);
        CHECK(hipMemcpy(d.work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
        for (int which : {0, 0, 1, 1}) PCHECK(launch(which, c, d, (int64_t)work.size() / 4));
        CHECK(hipDeviceSynchronize());
        printf("pmc mode: dispatches in order fp32 x2, bf16 x3 x2 (%d tiles, 6 per piece)\n", ntiles);
        return 0;
    }
    printf("device %s, %d CUs; %d tiles, fill %.2f\n", prop.gcnArchName, prop.multiProcessorCount, ntiles, fill);
    const char *names[2] = {"fp32 MFMA (pgcn_spmm_dense_f32)      ", "bf16 x3  (pgcn_spmm_dense_bf16x3_f32)"};
    const int real[2] = {0, 1};
    {   // ---- main case: Reddit-sized operand, f = 128 ------------------------------------------------------------
        Case c; make_case(c, 232965, 128, 128, ntiles, fill, 7, true);
        Dev d; upload(c, d, (int64_t)ntiles * 128);
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int per : {6, 3, 12}) {
            const std::vector<int32_t> work = make_work(ntiles, per);
            const int64_t np = (int64_t)work.size() / 4;
            CHECK(hipMemcpy(d.work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
            std::vector<int> sample = {0, 1, 2, (int)np / 2, (int)np - 2, (int)np - 1};
            for (int wi = 0; wi < 2; ++wi) {
                const int which = real[wi];
                for (int i = 0; i < 3; ++i) PCHECK(launch(which, c, d, np));
                CHECK(hipDeviceSynchronize());
                const int reps = 20;
                CHECK(hipEventRecord(e0, nullptr));
                for (int i = 0; i < reps; ++i) PCHECK(launch(which, c, d, np));
                CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double us = 1e3 * ms / reps;
                std::vector<float> got((size_t)np * 128 * c.f), again;
                CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
                PCHECK(launch(which, c, d, np)); CHECK(hipDeviceSynchronize());
                again.resize(got.size());
                CHECK(hipMemcpy(again.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
                const bool same = memcmp(got.data(), again.data(), got.size() * 4) == 0;
                printf("%s  %2d tiles/piece (%5lld pieces): %8.1f us per launch, %6.2f us per tile and CU, max err / sum|a||h| %.3e, deterministic %s\n",
                       names[which], per, (long long)np, us, us * prop.multiProcessorCount / ntiles, ref_err(c, work, got, sample), same ? "yes" : "NO");
                fflush(stdout);
            }
        }
        {   // ---- where a tile's time goes: the loop without panel staging, and without the A loads too (timing only) ----
            const std::vector<int32_t> work = make_work(ntiles, 6);
            const int64_t np = (int64_t)work.size() / 4;
            CHECK(hipMemcpy(d.work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
            const char *pn[2] = {"bf16 x3 without panel staging in the loop   ", "bf16 x3 without panel staging and A loads   "};
            for (int which = 2; which < 4; ++which) {
                for (int i = 0; i < 3; ++i) PCHECK(launch(which, c, d, np));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0, nullptr));
                for (int i = 0; i < 20; ++i) PCHECK(launch(which, c, d, np));
                CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("%s 6 tiles/piece: %8.1f us per launch, %6.2f us per tile and CU (timing only)\n", pn[which - 2], 1e3 * ms / 20,
                       1e3 * ms / 20 * prop.multiProcessorCount / ntiles);
            }
        }
        {   // ---- per-wave phase timers of the real kernel (s_memtime ticks, summed over the 24 quarters of a 6-tile piece) ----
            const std::vector<int32_t> work = make_work(ntiles, 6);
            const int64_t np = (int64_t)work.size() / 4;
            CHECK(hipMemcpy(d.work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
            unsigned long long *tb = nullptr;
            CHECK(hipMalloc(&tb, (size_t)np * 4 * 6 * 8));
            CHECK(hipMemset(tb, 0, (size_t)np * 4 * 6 * 8));
            PCHECK(pgcn_dense3_set_timers(tb));
            for (int i = 0; i < 3; ++i) PCHECK(launch(5, c, d, np));
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> t((size_t)np * 4 * 6);
            CHECK(hipMemcpy(t.data(), tb, t.size() * 8, hipMemcpyDeviceToHost));
            double m[6] = {0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < t.size(); ++i) m[i % 6] += (double)t[i];
            const double nw = (double)np * 4, nqt = 24.0;
            printf("phase timers (ticks per wave; per quarter = / 24): requests issued %.0f (%.0f), barrier wait %.0f (%.0f), MFMA block %.0f (%.0f), "
                   "split + LDS writes %.0f (%.0f), prologue %.0f, whole loop + prologue %.0f; 48 MFMAs of a quarter = 1536 pipe cycles\n",
                   m[0] / nw, m[0] / nw / nqt, m[1] / nw, m[1] / nw / nqt, m[2] / nw, m[2] / nw / nqt, m[3] / nw, m[3] / nw / nqt, m[4] / nw, m[5] / nw);
            PCHECK(pgcn_dense3_set_timers(nullptr));
            CHECK(hipFree(tb));
        }
        // ---- an Inf in an operand row that only structural zeros touch: the exact path must keep it out -----------
        {
            const std::vector<int32_t> work = make_work(ntiles, 6);
            CHECK(hipMemcpy(d.work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
            const int kz = 5;
            const int t = (int64_t)c.panel[0] * 128 + kz < c.ncols ? 0 : 2;     // (not on the partial last panel)
            Case c2 = c;                                  // (copy: tiles of piece 0 get a zero column, B an Inf row there)
            for (int i = 0; i < 128; ++i) c2.tiles[((size_t)t * 128 + i) * 128 + kz] = 0.f;
            const int64_t r = (int64_t)c2.panel[t] * 128 + kz;
            for (int tt = 0; tt < ntiles; ++tt) if (c2.panel[tt] == c2.panel[t]) for (int i = 0; i < 128; ++i) c2.tiles[((size_t)tt * 128 + i) * 128 + kz] = 0.f;
            for (int n = 0; n < c2.f; ++n) c2.B[(size_t)r * c2.ldb + n] = INFINITY;
            layouts(c2);
            CHECK(hipMemcpy(d.vals, c2.vals.data(), c2.vals.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d.planes, c2.planes.data(), c2.planes.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d.B, c2.B.data(), c2.B.size() * 4, hipMemcpyHostToDevice));
            const int64_t np = (int64_t)work.size() / 4;
            for (auto &x : c2.B) if (std::isinf(x)) x = 0.f;   // reference: the row is never referenced
            for (int wi = 0; wi < 2; ++wi) {
                const int which = real[wi];
                PCHECK(launch(which, c2, d, np)); CHECK(hipDeviceSynchronize());
                std::vector<float> got((size_t)np * 128 * c2.f);
                CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
                printf("%s  Inf row under structural zeros: piece 0 max err %.3e (exact path), piece 1 %.3e\n", names[which],
                       ref_err(c2, work, got, {0}), ref_err(c2, work, got, {1}));
            }
        }
        release(d);
    }
    {   // ---- f = 72, ldb = 75 (nothing aligned), small operand whose last panel holds 37 rows -------------------------
        Case c; make_case(c, 128 * 9 + 37, 72, 75, 64, 0.4, 11, true);
        Dev d; upload(c, d, 64 * 128);
        const std::vector<int32_t> work = make_work(64, 5);
        const int64_t np = (int64_t)work.size() / 4;
        CHECK(hipMemcpy(d.work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
        std::vector<int> all; for (int p = 0; p < np; ++p) all.push_back(p);
        for (int wi = 0; wi < 2; ++wi) {
            const int which = real[wi];
            PCHECK(launch(which, c, d, np)); CHECK(hipDeviceSynchronize());
            std::vector<float> got((size_t)np * 128 * c.f);
            CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
            printf("%s  f = 72, ldb = 75, partial last panel, all %lld pieces: max err / sum|a||h| %.3e\n", names[which], (long long)np,
                   ref_err(c, work, got, all));
        }
        release(d);
    }
    {   // ---- f = 200 (two feature blocks per piece: 128 + 72) ------------------------------------------------------------
        Case c; make_case(c, 4096, 200, 200, 48, 0.3, 13, false);
        Dev d; upload(c, d, 48 * 128);
        const std::vector<int32_t> work = make_work(48, 4);
        const int64_t np = (int64_t)work.size() / 4;
        CHECK(hipMemcpy(d.work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
        std::vector<int> all; for (int p = 0; p < np; ++p) all.push_back(p);
        for (int wi = 0; wi < 2; ++wi) {
            const int which = real[wi];
            PCHECK(launch(which, c, d, np)); CHECK(hipDeviceSynchronize());
            std::vector<float> got((size_t)np * 128 * c.f);
            CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
            printf("%s  f = 200: max err / sum|a||h| %.3e\n", names[which], ref_err(c, work, got, all));
        }
        release(d);
    }
    return 0;
}
