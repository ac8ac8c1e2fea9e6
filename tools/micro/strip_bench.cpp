// Stand-alone harness (r02) of pgcn_spmm_strip_f32 on synthetic strip records shaped like the benchmark graph's
// (1 024 pieces, ~60 k records, runs of ~3.2 records per panel -- every panel random, the worst case for the
// staging side -- 72 % of the pair slots used): time, clocks per record and CU, a sample of rows against a plain
// loop, and with PGCN_STRIP_PROBE=4 the per-wave phase timers.  A pure HIP binary: a gpurun call costs ~30 s.
// The second-generation kernel was developed on it next to the first one (both built in, records in both
// layouts): bit-identical partial sums, 6 560 -> 5 100 clk per record; profiles/r02_strip_bench.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>

static const int32_t PADOFF = 128 * 512;      // PGCN_STRIP pad slot: byte offset of the all-zero LDS row (partition.STRIP_PAD_OFF)

#ifdef WITH_HALF   // tools/experiments/pgcn_spmm_strip_half.hip (r02), not part of libpgcn_hip.so: link it in to use mode 3
extern "C" int pgcn_spmm_strip_half_f32(const int32_t *, int64_t, const int32_t *, const int32_t *, const float *, int64_t, int64_t,
                                        int32_t, float *, int64_t, int64_t, void *);
#endif
#ifdef WITH_NEXT
extern "C" int pgcn_spmm_strip3_f32(const int32_t *, int64_t, const int32_t *, const int32_t *, const float *, int64_t, int64_t,
                                    int32_t, float *, int64_t, int64_t, void *);
#endif
extern "C" int pgcn_spmm_strip_f32(const int32_t *, int64_t, const int32_t *, const int32_t *, const float *, int64_t, int64_t,
                                    int32_t, float *, int64_t, int64_t, void *);
extern "C" const char *pgcn_last_error(void);

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int64_t n = 232965;
    const int f = argc > 1 ? atoi(argv[1]) : 128;
    const int npieces = argc > 2 ? atoi(argv[2]) : 1024;
    const double fill = 0.72;
    std::mt19937 rng(7);
    std::vector<int32_t> work, recs;
    std::vector<int32_t> pairs_new, pairs_next;
    int64_t nrec = 0;
    const int npanels = (int)(n / 128);   // full panels only (v2 windows the last partial panel differently from v1)
    for (int p = 0; p < npieces; ++p) {
        const int R = getenv("STRIP_BENCH_R") ? atoi(getenv("STRIP_BENCH_R")) : 40 + (int)(rng() % 40);
        work.insert(work.end(), {p % 455, (int32_t)nrec, (int32_t)(nrec + R), p * 512});
        std::vector<int> panel(R), flag(R);
        int k = 0;
        while (k < R) {
            int len = 1;
            while (len < 12 && (rng() % 1000) < 690) ++len;      // geometric, mean ~3.2
            const int pn = (int)(rng() % npanels);
            for (int i = 0; i < len && k < R; ++i, ++k) { panel[k] = pn; flag[k] = i > 0; }
        }
        for (k = 0; k < R; ++k) {
            int nextp = -1, next2 = -1;   // panels of the piece's next run and of the one after it (the latter only read by the
                                          // three-buffer experiment of the half-footprint kernel, profiles/r02_strip_bench.txt)
            if (!flag[k]) {
                int q = k + 1; while (q < R && flag[q]) ++q;
                if (q < R) { nextp = panel[q]; ++q; while (q < R && flag[q]) ++q; if (q < R) next2 = panel[q]; }
            }
            recs.insert(recs.end(), {panel[k], flag[k], nextp, next2});
        }
        nrec += R;
    }
    pairs_new.resize((size_t)nrec * 2048); pairs_next.resize((size_t)nrec * 2048);
    for (int64_t r = 0; r < nrec; ++r) {
        const int panel = recs[r * 4];
        const int rows_in_panel = (int)std::min<int64_t>(getenv("STRIP_BENCH_ROWS") ? atoi(getenv("STRIP_BENCH_ROWS")) : 128, n - (int64_t)panel * 128);
        for (int row = 0; row < 512; ++row)
            for (int u = 0; u < 2; ++u) {
                int32_t off = PADOFF; float v = 0.f;
                if ((rng() % 1000) < fill * 1000) { off = (int32_t)(rng() % rows_in_panel) * 512; v = (float)(rng() % 2001) / 1000.f - 1.f; }
                int32_t vb; memcpy(&vb, &v, 4);
                const size_t in = (size_t)r * 2048 + (((row % 64) * 8 + row / 64) * 2 + u) * 2;
                pairs_new[in] = off; pairs_new[in + 1] = vb;
                // next-generation format: 256-byte rows, unused slot = {0, -0.0f}
                pairs_next[in] = off == PADOFF ? 0 : off / 2; pairs_next[in + 1] = off == PADOFF ? (int32_t)0x80000000 : vb;
            }
    }
    // ... and the header of record k + 1 in the low bytes of the four offsets of every group's row slots 0 and 1
    for (int p = 0; p < npieces; ++p)
        for (int64_t k = work[p * 4 + 1]; k < work[p * 4 + 2]; ++k) {
            uint32_t hu = 0;
            if (k + 1 < work[p * 4 + 2]) hu = (uint32_t)recs[(k + 1) * 4 + 1] | ((uint32_t)(recs[(k + 1) * 4 + 2] + 1) << 1);
            for (int g = 0; g < 64; ++g)
                for (int b = 0; b < 4; ++b) pairs_next[(size_t)k * 2048 + ((g * 8 + b / 2) * 2 + b % 2) * 2] |= (int32_t)((hu >> (8 * b)) & 255u);
        }
    std::vector<float> hB((size_t)n * f);
    for (auto &x : hB) x = (float)(rng() % 2001) / 1000.f - 1.f;
    int32_t *dwork, *drecs, *dpn, *dpx; float *dB, *dws_n, *dws_x;
    const int64_t nslots = (int64_t)npieces * 512;
    CHECK(hipMalloc(&dwork, work.size() * 4)); CHECK(hipMemcpy(dwork, work.data(), work.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&drecs, recs.size() * 4)); CHECK(hipMemcpy(drecs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dpn, pairs_new.size() * 4)); CHECK(hipMemcpy(dpn, pairs_new.data(), pairs_new.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dB, hB.size() * 4)); CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dws_n, nslots * f * 4)); CHECK(hipMalloc(&dws_x, nslots * f * 4)); CHECK(hipMemset(dws_x, 0xdd, nslots * f * 4));
    CHECK(hipMalloc(&dpx, pairs_next.size() * 4)); CHECK(hipMemcpy(dpx, pairs_next.data(), pairs_next.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dws_n, 0xee, nslots * f * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, int which) {
        float best = 1e9f, sum = 0.f;
        const int reps = 6;
        for (int it = 0; it < reps + 1; ++it) {
            CHECK(hipEventRecord(e0));
            int rc;
#ifdef WITH_HALF
            if (which == 3) rc = pgcn_spmm_strip_half_f32(dwork, npieces, drecs, dpn, dB, f, n, f, dws_x, nslots * f, nslots, nullptr);
            else
#endif
#ifdef WITH_NEXT
            if (which == 2) rc = pgcn_spmm_strip3_f32(dwork, npieces, drecs, dpx, dB, f, n, f, dws_x, nslots * f, nslots, nullptr);
            else
#endif
            rc = pgcn_spmm_strip_f32(dwork, npieces, drecs, dpn, dB, f, n, f, dws_n, nslots * f, nslots, nullptr);
            if (rc) { printf("%s failed: %s\n", name, pgcn_last_error()); exit(1); }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it) { best = ms < best ? ms : best; sum += ms; }
        }
        const double clk = (sum / reps) * 1e-3 * 2.4e9 * 256 / (double)nrec;
        printf("%-10s f=%d pieces=%d records=%lld : mean %.3f ms  best %.3f ms  = %.0f clk per record and CU\n", name, f, npieces,
               (long long)nrec, sum / reps, best, clk);
    };
    timeit("strip", 1);
#ifdef WITH_HALF
    timeit("strip half", 3);
#endif
    if (!(getenv("PGCN_STRIP_PROBE") && atoi(getenv("PGCN_STRIP_PROBE")))) {
        std::vector<float> ha((size_t)nslots * f), hb((size_t)nslots * f);
        CHECK(hipMemcpy(ha.data(), dws_n, ha.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hb.data(), dws_x, hb.size() * 4, hipMemcpyDeviceToHost));
        size_t nd = 0;
        for (size_t i = 0; i < ha.size(); ++i)
            if (memcmp(&ha[i], &hb[i], 4)) { if (nd < 5) printf("  mismatch at slot %zu col %zu: %g vs %g\n", i / f, i % f, ha[i], hb[i]); ++nd; }
        printf("strip vs strip half: %zu differing values of %zu\n", nd, ha.size());
    }
#ifdef WITH_NEXT
    timeit("strip next", 2);
    if (!(getenv("PGCN_STRIP_PROBE") && atoi(getenv("PGCN_STRIP_PROBE")))) {
        std::vector<float> ha((size_t)nslots * f), hb((size_t)nslots * f);
        CHECK(hipMemcpy(ha.data(), dws_n, ha.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hb.data(), dws_x, hb.size() * 4, hipMemcpyDeviceToHost));
        size_t nd = 0;
        for (size_t i = 0; i < ha.size(); ++i)
            if (memcmp(&ha[i], &hb[i], 4)) { if (nd < 5) printf("  mismatch at slot %zu col %zu: %g vs %g\n", i / f, i % f, ha[i], hb[i]); ++nd; }
        printf("current vs next: %zu differing values of %zu\n", nd, ha.size());
    }
    CHECK(hipMemcpy(dws_n, dws_x, (size_t)nslots * f * 4, hipMemcpyDeviceToDevice));   // the checks below look at the next kernel
#endif
    std::vector<float> hn((size_t)nslots * f);
    CHECK(hipMemcpy(hn.data(), dws_n, hn.size() * 4, hipMemcpyDeviceToHost));
    if (getenv("PGCN_STRIP_PROBE") && (atoi(getenv("PGCN_STRIP_PROBE")) & 4)) {
        double t[4] = {0, 0, 0, 0};
        for (int p = 0; p < npieces; ++p)
            for (int w = 0; w < 16; ++w)
                for (int i = 0; i < 4; ++i) t[i] += hn[((size_t)work[p * 4 + 3] + w) * f + i];
        printf("timer ticks per record and wave: vm wait %.1f  header %.1f  barrier+panel issue %.1f  compute %.1f\n",
               t[0] / npieces / 16, t[1] / npieces / 16, t[2] / npieces / 16, t[3] / npieces / 16);
        printf("per wave (vm / barrier / compute):");
        for (int w = 0; w < 16; ++w) {
            double a = 0, b = 0, c = 0;
            for (int p = 0; p < npieces; ++p) { const float *o = &hn[((size_t)work[p * 4 + 3] + w) * f]; a += o[0]; b += o[2]; c += o[3]; }
            printf(" w%d %.0f/%.0f/%.0f", w, a / npieces, b / npieces, c / npieces);
        }
        printf("\n");
        return 0;
    }
    size_t bad = 0;
    // sample of rows / features of a few pieces against a plain loop (same fmaf order: expected exact)
    double maxref = 0;
    for (int pc = 0; pc < npieces; pc += 97) {
        const int k0 = work[pc * 4 + 1], k1 = work[pc * 4 + 2];
        for (int row = 0; row < 512; row += 37)
            for (int c = 0; c < f; c += 5) {
                float acc = 0.f;
                for (int k = k0; k < k1; ++k)
                    for (int u = 0; u < 2; ++u) {
                        const size_t in = (size_t)k * 2048 + (((row % 64) * 8 + row / 64) * 2 + u) * 2;
                        const int32_t off = pairs_new[in];
                        if (off == PADOFF) continue;
                        float v; memcpy(&v, &pairs_new[in + 1], 4);
                        acc = fmaf(v, hB[((size_t)recs[k * 4] * 128 + (off >> 9)) * f + c], acc);
                    }
                const double d = fabs((double)acc - hn[(size_t)(work[pc * 4 + 3] + row) * f + c]); if (d > maxref) maxref = d;
                if (d != 0) ++bad;
            }
    }
    printf("kernel vs plain loop (sample): %zu differing values, max |diff| %g\n", bad, maxref);
    return bad ? 2 : 0;
}
