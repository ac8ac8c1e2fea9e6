// Stand-alone harness of pgcn_spmm_dense_bf16x3_f32 (r04: 512 x 128 blocks, panels split once per SpMM into a
// work-space, A kept fp32 and split in registers) next to pgcn_spmm_dense_f32 on the same data (every block = four
// stacked 128 x 128 tiles of the fp32 kernel; "tile" below always means 128 x 128), a Reddit-sized operand (232 965 x 128).  Prints the time per launch and per tile and CU of both kernels, the cost of the panel
// split alone, the error of both against a float64 loop relative to sum |a||h| per output, run-to-run determinism,
// and the edge cases (f = 72 with an odd leading dimension, the last partial panel, f = 200, an Inf in an operand row
// that only structural zeros touch).  A pure HIP binary: no torch, a gpurun call costs ~20 s.
//   build: tools/micro/build_dense3_bench.sh (links lib/libpgcn_hip.so; the binary travels with the snapshot)
//   usage: dense3_bench.bin [nblocks=768] [fill=0.3] [distinct panels=1820]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

extern "C" int pgcn_spmm_dense_f32(const int32_t *, int64_t, const int32_t *, const float *, const float *, int64_t, int64_t,
                                   int32_t, float *, int64_t, int64_t, void *);
extern "C" int pgcn_spmm_dense_bf16x3_f32(const int32_t *, int64_t, const int32_t *, const float *, const int32_t *, int64_t,
                                          const float *, int64_t, int64_t, int32_t, void *, int64_t, float *, int64_t, int64_t, void *);
extern "C" int64_t pgcn_dense_bf16x3_image_bytes(int64_t, int32_t);
// the kernel source compiled into this binary again with PGCN_DENSE3_PROBE = 1..5 (see the kernel file)
#define PROBE_DECL(N) extern "C" int pgcn_spmm_dense_bf16x3_probe##N##_f32(const int32_t *, int64_t, const int32_t *, const float *, const int32_t *, int64_t, \
                                          const float *, int64_t, int64_t, int32_t, void *, int64_t, float *, int64_t, int64_t, void *);
PROBE_DECL(1) PROBE_DECL(5)
extern "C" const char *pgcn_last_error(void);

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define PCHECK(x) do { int r_ = (x); if (r_ != 0) { printf("pgcn error %d (%s) at line %d\n", r_, pgcn_last_error(), __LINE__); exit(1); } } while (0)

struct Case {
    int64_t ncols; int f; int64_t ldb; int nblocks;
    std::vector<float> blocks;         // [nblocks][512][128] row-major (i, k)
    std::vector<int32_t> panel;        // [nblocks] column block of every block
    std::vector<int32_t> panel4;       // [4][nblocks] the same per 128 x 128 tile, tile = sub * nblocks + block (fp32 kernel)
    std::vector<float> B;              // [ncols][ldb]
    std::vector<float> vals, vals3;    // fp32-MFMA swizzle (tile order) / A-operand order of the bf16 MFMA (block order), both fp32
    std::vector<int32_t> plist, bimg;  // distinct panels (sorted), block -> index in plist
};

static void layouts(Case &c) {
    const int nb = c.nblocks;
    c.vals.assign((size_t)nb * 65536, 0.f);
    c.vals3.assign((size_t)nb * 65536, 0.f);
    c.panel4.resize((size_t)4 * nb);
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 512; ++i)
            for (int k = 0; k < 128; ++k) {
                const float a = c.blocks[((size_t)b * 512 + i) * 128 + k];
                {   // fp32 kernel: tile (sub, b), row i % 128
                    const int sub = i / 128, it = i % 128, w = it / 32, il = it % 32, s = k / 2, kh = k % 2;
                    const size_t t = (size_t)sub * nb + b;
                    c.panel4[t] = c.panel[b];
                    c.vals[t * 16384 + ((w * 16 + s / 4) * 64 + kh * 32 + il) * 4 + s % 4] = a;
                }
                const int w = i / 64, rb = (i / 32) % 2, il = i % 32, ks = k / 16, hk = (k / 8) % 2, h = (k / 4) % 2, e = k % 4;
                c.vals3[(size_t)b * 65536 + ((((((size_t)w * 16 + 2 * ks + rb) * 2 + h) * 64) + hk * 32 + il) * 4) + e] = a;
            }
    c.plist = c.panel;
    std::sort(c.plist.begin(), c.plist.end());
    c.plist.erase(std::unique(c.plist.begin(), c.plist.end()), c.plist.end());
    c.bimg.resize(nb);
    for (int b = 0; b < nb; ++b)
        c.bimg[b] = (int32_t)(std::lower_bound(c.plist.begin(), c.plist.end(), c.panel[b]) - c.plist.begin());
}

static void make_case(Case &c, int64_t ncols, int f, int64_t ldb, int nblocks, double fill, unsigned seed, bool last_panel_first, int distinct) {
    c.ncols = ncols; c.f = f; c.ldb = ldb; c.nblocks = nblocks;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const int npanels = std::min((int)((ncols + 127) / 128), std::max(distinct, 1));
    c.panel.resize(nblocks);
    c.blocks.assign((size_t)nblocks * 65536, 0.f);
    for (int b = 0; b < nblocks; ++b) {
        c.panel[b] = (int32_t)(rng() % npanels);
        if (last_panel_first && b == 1) c.panel[b] = (int32_t)((ncols + 127) / 128) - 1;   // the partial panel: rows >= ncols read as zero
        const int64_t valid = std::min<int64_t>(128, ncols - (int64_t)c.panel[b] * 128);
        for (int i = 0; i < 512; ++i)
            for (int k = 0; k < valid; ++k)
                if (U(rng) < fill) c.blocks[((size_t)b * 512 + i) * 128 + k] = (0.01f + U(rng)) * 0.05f * (rng() & 1 ? 1.f : -1.f);
    }
    c.B.resize((size_t)ncols * ldb);
    for (auto &x : c.B) x = 2.f * U(rng) - 1.f;
    layouts(c);
}

struct Dev { int32_t *work, *work4, *panel4, *plist, *bimg; float *vals, *vals3, *B, *ws; void *img; int64_t img_bytes; };

// work: block pieces {block row, first block, blocks, first slot (512 per piece)}; errors of piece p over its 512 rows
static double ref_err(const Case &c, const std::vector<int32_t> &work, const std::vector<float> &got, const std::vector<int> &pieces) {
    double worst = 0;
    for (int p : pieces) {
        const int b0 = work[4 * p + 1], nb = work[4 * p + 2], slot = work[4 * p + 3];
        for (int i = 0; i < 512; ++i)
            for (int n = 0; n < c.f; ++n) {
                double s = 0, sa = 0;
                for (int b = b0; b < b0 + nb; ++b)
                    for (int k = 0; k < 128; ++k) {
                        const float a = c.blocks[((size_t)b * 512 + i) * 128 + k];
                        if (a == 0.f) continue;
                        const double h = c.B[((size_t)c.panel[b] * 128 + k) * c.ldb + n];
                        s += (double)a * h; sa += std::fabs((double)a * h);
                    }
                const double g = got[((size_t)slot + i) * c.f + n];
                const double e = std::isfinite(g) ? std::fabs(g - s) / (sa > 0 ? sa : 1.0) : 1e30;
                if (e > worst) worst = e;
            }
    }
    return worst;
}

// block pieces of `per` blocks, and the same work for the fp32 kernel: four 128-row pieces per block piece (tiles
// (sub, b0..) are contiguous in its tile order), writing the same 512-row slot block
static void make_work(int nblocks, int per, std::vector<int32_t> &w, std::vector<int32_t> &w4) {
    w.clear(); w4.clear();
    int p = 0;
    for (int b = 0; b < nblocks; b += per, ++p) {
        const int n = std::min(per, nblocks - b);
        w.push_back(p % 455); w.push_back(b); w.push_back(n); w.push_back(p * 512);
        for (int sub = 0; sub < 4; ++sub) { w4.push_back((p % 455) * 4 + sub); w4.push_back(sub * nblocks + b); w4.push_back(n); w4.push_back(p * 512 + sub * 128); }
    }
}

template <class T> static T *up(const std::vector<T> &v) {
    T *d = nullptr;
    CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
static void upload(const Case &c, Dev &d, int64_t max_slots) {
    d.panel4 = up(c.panel4); d.plist = up(c.plist); d.bimg = up(c.bimg); d.vals = up(c.vals); d.vals3 = up(c.vals3); d.B = up(c.B);
    d.img_bytes = pgcn_dense_bf16x3_image_bytes((int64_t)c.plist.size(), c.f);
    CHECK(hipMalloc(&d.img, (size_t)d.img_bytes));
    CHECK(hipMalloc(&d.ws, (size_t)max_slots * c.f * 4));
    CHECK(hipMalloc(&d.work, (size_t)c.nblocks * 16));
    CHECK(hipMalloc(&d.work4, (size_t)c.nblocks * 64));
}
static void set_work(Dev &d, const std::vector<int32_t> &w, const std::vector<int32_t> &w4) {
    CHECK(hipMemcpy(d.work, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d.work4, w4.data(), w4.size() * 4, hipMemcpyHostToDevice));
}
static void release(Dev &d) { hipFree(d.panel4); hipFree(d.plist); hipFree(d.bimg); hipFree(d.vals); hipFree(d.vals3); hipFree(d.B); hipFree(d.ws); hipFree(d.work); hipFree(d.work4); hipFree(d.img); }

static int launch(int which, const Case &c, const Dev &d, int64_t npieces) {
    const int64_t ns = npieces * 512;
    if (which == 0) return pgcn_spmm_dense_f32(d.work4, npieces * 4, d.panel4, d.vals, d.B, c.ldb, c.ncols, c.f, d.ws, ns * c.f, ns, nullptr);
    auto fn = pgcn_spmm_dense_bf16x3_f32;
    if (which == 11) fn = pgcn_spmm_dense_bf16x3_probe1_f32;
    if (which == 15) fn = pgcn_spmm_dense_bf16x3_probe5_f32;
    return fn(d.work, npieces, d.bimg, d.vals3, d.plist, (int64_t)c.plist.size(), d.B, c.ldb, c.ncols, c.f, d.img,
              d.img_bytes, d.ws, ns * c.f, ns, nullptr);
}

static double time_us(int which, const Case &c, const Dev &d, int64_t np, int reps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) PCHECK(launch(which, c, d, np));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) PCHECK(launch(which, c, d, np));
    CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return 1e3 * ms / reps;
}

int main(int argc, char **argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 768;
    const double fill = argc > 2 ? atof(argv[2]) : 0.3;
    const int distinct = argc > 3 ? atoi(argv[3]) : 1820;
    const int ntiles = nblocks * 4;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    std::vector<int32_t> work, work4;
    if (getenv("DENSE3_PMC")) {            // under rocprofv3: two launches each of the fp32 and the bf16 kernel, 2 blocks per piece
        Case c; make_case(c, 232965, 128, 128, nblocks, fill, 7, true, distinct);
        Dev d; upload(c, d, (int64_t)nblocks * 512);
        make_work(nblocks, 2, work, work4); set_work(d, work, work4);
        for (int which : {0, 0, 1, 1}) PCHECK(launch(which, c, d, (int64_t)work.size() / 4));
        CHECK(hipDeviceSynchronize());
        printf("pmc mode: dispatches in order fp32 x2, (split + bf16 x3) x2 (%d blocks, 2 per piece)\n", nblocks);
        return 0;
    }
    printf("device %s, %d CUs; %d blocks = %d tiles, fill %.2f, %d distinct panels at most\n", prop.gcnArchName, prop.multiProcessorCount, nblocks, ntiles, fill, distinct);
    const char *names[2] = {"fp32 MFMA (pgcn_spmm_dense_f32)      ", "bf16 x3  (pgcn_spmm_dense_bf16x3_f32)"};
    {   // ---- main case: Reddit-sized operand, f = 128 ------------------------------------------------------------
        Case c; make_case(c, 232965, 128, 128, nblocks, fill, 7, true, distinct);
        Dev d; upload(c, d, (int64_t)nblocks * 512);
        printf("%zu distinct panels: image work-space %.1f MB\n", c.plist.size(), d.img_bytes / 1048576.0);
        for (int per : {2, 1, 3, 4, 8}) {
            make_work(nblocks, per, work, work4); set_work(d, work, work4);
            const int64_t np = (int64_t)work.size() / 4;
            std::vector<int> sample = {0, 1, (int)np / 2, (int)np - 1};
            for (int which = 0; which < 2; ++which) {
                const double us = time_us(which, c, d, np, 20);
                std::vector<float> got((size_t)np * 512 * c.f), again;
                CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
                PCHECK(launch(which, c, d, np)); CHECK(hipDeviceSynchronize());
                again.resize(got.size());
                CHECK(hipMemcpy(again.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
                const bool same = memcmp(got.data(), again.data(), got.size() * 4) == 0;
                printf("%s  %2d blocks/piece (%5lld pieces): %8.1f us per launch, %6.2f us per tile and CU, max err / sum|a||h| %.3e, deterministic %s\n",
                       names[which], per, (long long)np, us, us * prop.multiProcessorCount / ntiles, ref_err(c, work, got, sample), same ? "yes" : "NO");
                fflush(stdout);
            }
        }
        for (int per : {4, 8}) {   // ---- where the time goes: timing-only variants and the phase timers ----
            make_work(nblocks, per, work, work4); set_work(d, work, work4);
            const int64_t np = (int64_t)work.size() / 4;
            {
                const double us = time_us(11, c, d, np, 10);
                printf("probe (%d blocks/piece) no MFMAs (copies, LDS reads, barriers only): %8.1f us per launch, %6.2f us per tile and CU (timing only)\n", per, us,
                       us * prop.multiProcessorCount / ntiles);
            }
            const double us = time_us(15, c, d, np, 5);
            std::vector<float> got((size_t)np * 512 * c.f);
            CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
            double top = 0, cmp = 0, nq = 0, tot = 0; int64_t nw = 0;
            for (int64_t p = 0; p < np; ++p)
                for (int w = 0; w < 8; ++w, ++nw) {
                    const float *o = &got[((size_t)work[4 * p + 3] + 64 * w) * c.f];
                    top += o[0]; cmp += o[(size_t)c.f]; nq += o[(size_t)2 * c.f]; tot += o[(size_t)3 * c.f];
                }
            printf("phase timers (%d blocks/piece, %.1f us per launch): per quarter and wave %.0f ticks waiting at the top (copies + barrier), %.0f computing "
                   "(192 MFMAs of a SIMD's two waves = 6144 pipe cycles); whole loop %.0f ticks per wave = %.2f GHz if ticks are shader cycles\n",
                   per, us, top / nq, cmp / nq, tot / nw, tot / nw / (us - 30.0) * 1e-3);
            fflush(stdout);
        }
        {   // the split alone: the same call with ONE one-block piece (the split covers all listed panels whatever the work list holds)
            make_work(1, 1, work, work4); set_work(d, work, work4);
            printf("panel split + one block: %8.1f us per launch (%zu panels)\n", time_us(1, c, d, 1, 20), c.plist.size());
        }
        // ---- an Inf in an operand row that only structural zeros touch: the exact path must keep it out -----------
        {
            make_work(nblocks, 2, work, work4); set_work(d, work, work4);
            const int kz = 5;
            const int b = (int64_t)c.panel[0] * 128 + kz < c.ncols ? 0 : 2;     // (not on the partial last panel)
            Case c2 = c;                                  // (copy: the blocks of that panel get a zero column, B an Inf row there)
            const int64_t r = (int64_t)c2.panel[b] * 128 + kz;
            for (int bb = 0; bb < nblocks; ++bb) if (c2.panel[bb] == c2.panel[b]) for (int i = 0; i < 512; ++i) c2.blocks[((size_t)bb * 512 + i) * 128 + kz] = 0.f;
            for (int n = 0; n < c2.f; ++n) c2.B[(size_t)r * c2.ldb + n] = INFINITY;
            layouts(c2);
            CHECK(hipMemcpy(d.vals, c2.vals.data(), c2.vals.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d.vals3, c2.vals3.data(), c2.vals3.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d.B, c2.B.data(), c2.B.size() * 4, hipMemcpyHostToDevice));
            const int64_t np = (int64_t)work.size() / 4;
            for (auto &x : c2.B) if (std::isinf(x)) x = 0.f;   // reference: the row is never referenced
            for (int which = 0; which < 2; ++which) {
                PCHECK(launch(which, c2, d, np)); CHECK(hipDeviceSynchronize());
                std::vector<float> got((size_t)np * 512 * c2.f);
                CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
                printf("%s  Inf row under structural zeros: piece %d max err %.3e (exact path), piece %d %.3e\n", names[which], b / 2,
                       ref_err(c2, work, got, {b / 2}), b / 2 + 1, ref_err(c2, work, got, {b / 2 + 1}));
            }
        }
        release(d);
    }
    {   // ---- f = 72, ldb = 75 (nothing aligned), small operand whose last panel holds 37 rows -------------------------
        Case c; make_case(c, 128 * 9 + 37, 72, 75, 16, 0.4, 11, true, 1820);
        Dev d; upload(c, d, 16 * 512);
        make_work(16, 3, work, work4); set_work(d, work, work4);
        const int64_t np = (int64_t)work.size() / 4;
        std::vector<int> all; for (int p = 0; p < np; ++p) all.push_back(p);
        for (int which = 0; which < 2; ++which) {
            PCHECK(launch(which, c, d, np)); CHECK(hipDeviceSynchronize());
            std::vector<float> got((size_t)np * 512 * c.f);
            CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
            printf("%s  f = 72, ldb = 75, partial last panel, all %lld pieces: max err / sum|a||h| %.3e\n", names[which], (long long)np,
                   ref_err(c, work, got, all));
        }
        release(d);
    }
    for (int f : {200, 16, 40}) {   // ---- f = 200 (two feature blocks per piece: 128 + 72), f = 16 and 40 (one and two column blocks) ----
        Case c; make_case(c, 4096, f, f, 12, 0.3, 13, false, 1820);
        Dev d; upload(c, d, 12 * 512);
        make_work(12, 2, work, work4); set_work(d, work, work4);
        const int64_t np = (int64_t)work.size() / 4;
        std::vector<int> all; for (int p = 0; p < np; ++p) all.push_back(p);
        for (int which = 0; which < 2; ++which) {
            PCHECK(launch(which, c, d, np)); CHECK(hipDeviceSynchronize());
            std::vector<float> got((size_t)np * 512 * c.f);
            CHECK(hipMemcpy(got.data(), d.ws, got.size() * 4, hipMemcpyDeviceToHost));
            printf("%s  f = %d: max err / sum|a||h| %.3e\n", names[which], f, ref_err(c, work, got, all));
        }
        release(d);
    }
    return 0;
}
