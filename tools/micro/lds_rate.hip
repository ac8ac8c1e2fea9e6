// Micro-benchmark (r02): what bounds the compute phase of the strip kernel (pgcn_spmm_strip.hip)?
// One 1024-thread workgroup per CU, a 129-row x 512 B panel in LDS, every 32-lane group reads whole
// rows at random offsets -- the access pattern of the strip records -- in several instruction shapes:
//   A  the shipped step: 2 pair reads (b128, group-uniform address) prefetched one step ahead + 4 row
//      reads (b128) per wait, 8 steps per record
//   B  8 row reads + 4 pair reads per wait (twice the depth)
//   C  row reads only, 16 per wait, offsets held in registers (the ceiling of the row reads alone)
//   D  as C, the whole wave reads 1 KB of consecutive bytes (two adjacent rows)
//   E  ds_read_b64, one row per WAVE (64 lanes x 8 B), 16 per wait
//   F  as A but the pair reads are ds_read_b64 of one pair each (4 per step)
// Prints LDS bytes per clock and CU for the row reads (pairs on top), at 16 and 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kPanelBytes = 129 * 512;
constexpr int kPairBytes = 8192;       // 512 rows x 2 pairs x 8 B
constexpr int kSmem = kPanelBytes + kPairBytes;

template <int OFF>
__device__ __forceinline__ void rd128(f32x4 &v, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void rd64(f32x2 &v, uint32_t addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void w2(f32x4 &a, f32x4 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N>
__device__ __forceinline__ void w4(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void w4h(f32x2 &a, f32x2 &b, f32x2 &c, f32x2 &d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ void fma_row(f32x2 (&acc)[2], float w, const f32x4 &x) {
    const f32x2 ww = {w, w};
    acc[0] = __builtin_elementwise_fma(ww, f32x2{x.x, x.y}, acc[0]);
    acc[1] = __builtin_elementwise_fma(ww, f32x2{x.z, x.w}, acc[1]);
}

template <int VAR, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void lds_rate(const int2 *__restrict__ pairs_g, float4 *__restrict__ out, int records) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int sub = lane & 31;
    const int group = threadIdx.x >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    for (int i = threadIdx.x; i < kPanelBytes / 4; i += THREADS) reinterpret_cast<float *>(smem)[i] = 1.0f + (i & 7);
    for (int i = threadIdx.x; i < kPairBytes / 8; i += THREADS) reinterpret_cast<int2 *>(smem + kPanelBytes)[i] = pairs_g[i];
    __syncthreads();
    f32x2 acc[16][2];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j][0] = acc[j][1] = f32x2{0.f, 0.f};
    const uint32_t pa = lds0 + kPanelBytes + (group % (kPairBytes / 256)) * 256;
    const uint32_t rowbase = lds0 + sub * 16;

    if constexpr (VAR == 0) {          // A: the shipped step
        for (int r = 0; r < records; ++r) {
            f32x4 p0, p1, n0, n1;
            rd128<0>(p0, pa);
            rd128<16>(p1, pa);
#define STEP(J, MORE)                                                         \
            {                                                                 \
                w2<0>(p0, p1);                                                \
                f32x4 x0, x1, x2, x3;                                         \
                rd128<0>(x0, rowbase + (uint32_t)__float_as_int(p0.x));       \
                rd128<0>(x1, rowbase + (uint32_t)__float_as_int(p0.z));       \
                rd128<0>(x2, rowbase + (uint32_t)__float_as_int(p1.x));       \
                rd128<0>(x3, rowbase + (uint32_t)__float_as_int(p1.z));       \
                if (MORE) {                                                   \
                    rd128<((J) + 2) * 16>(n0, pa);                            \
                    rd128<((J) + 3) * 16>(n1, pa);                            \
                    w4<2>(x0, x1, x2, x3);                                    \
                } else {                                                      \
                    w4<0>(x0, x1, x2, x3);                                    \
                }                                                             \
                fma_row(acc[(J)], p0.y, x0);                                  \
                fma_row(acc[(J)], p0.w, x1);                                  \
                fma_row(acc[(J) + 1], p1.y, x2);                              \
                fma_row(acc[(J) + 1], p1.w, x3);                              \
                if (MORE) { p0 = n0; p1 = n1; }                               \
            }
            STEP(0, true) STEP(2, true) STEP(4, true) STEP(6, true) STEP(8, true) STEP(10, true) STEP(12, true) STEP(14, false)
#undef STEP
        }
    } else if constexpr (VAR == 1) {   // B: 8 rows + 4 pair reads per wait
        for (int r = 0; r < records; ++r) {
            f32x4 p0, p1, p2, p3, n0, n1, n2, n3;
            rd128<0>(p0, pa); rd128<16>(p1, pa); rd128<32>(p2, pa); rd128<48>(p3, pa);
#define STEP8(J, MORE)                                                        \
            {                                                                 \
                w4<0>(p0, p1, p2, p3);                                        \
                f32x4 x0, x1, x2, x3, x4, x5, x6, x7;                         \
                rd128<0>(x0, rowbase + (uint32_t)__float_as_int(p0.x));       \
                rd128<0>(x1, rowbase + (uint32_t)__float_as_int(p0.z));       \
                rd128<0>(x2, rowbase + (uint32_t)__float_as_int(p1.x));       \
                rd128<0>(x3, rowbase + (uint32_t)__float_as_int(p1.z));       \
                rd128<0>(x4, rowbase + (uint32_t)__float_as_int(p2.x));       \
                rd128<0>(x5, rowbase + (uint32_t)__float_as_int(p2.z));       \
                rd128<0>(x6, rowbase + (uint32_t)__float_as_int(p3.x));       \
                rd128<0>(x7, rowbase + (uint32_t)__float_as_int(p3.z));       \
                if (MORE) {                                                   \
                    rd128<((J) + 4) * 16>(n0, pa); rd128<((J) + 5) * 16>(n1, pa); \
                    rd128<((J) + 6) * 16>(n2, pa); rd128<((J) + 7) * 16>(n3, pa); \
                    w4<8>(x0, x1, x2, x3);                                    \
                } else {                                                      \
                    w4<4>(x0, x1, x2, x3);                                    \
                }                                                             \
                fma_row(acc[(J)], p0.y, x0);                                  \
                fma_row(acc[(J)], p0.w, x1);                                  \
                fma_row(acc[(J) + 1], p1.y, x2);                              \
                fma_row(acc[(J) + 1], p1.w, x3);                              \
                if (MORE) { w4<4>(x4, x5, x6, x7); } else { w4<0>(x4, x5, x6, x7); } \
                fma_row(acc[(J) + 2], p2.y, x4);                              \
                fma_row(acc[(J) + 2], p2.w, x5);                              \
                fma_row(acc[(J) + 3], p3.y, x6);                              \
                fma_row(acc[(J) + 3], p3.w, x7);                              \
                if (MORE) { p0 = n0; p1 = n1; p2 = n2; p3 = n3; }             \
            }
            STEP8(0, true) STEP8(4, true) STEP8(8, true) STEP8(12, false)
#undef STEP8
        }
    } else if constexpr (VAR == 2 || VAR == 3) {   // C / D: rows only, NR per wait, offsets in registers
        constexpr int NR = THREADS > 512 ? 8 : 16;
        uint32_t a[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int2 p = reinterpret_cast<const int2 *>(smem + kPanelBytes)[(group * 16 + j) % 1024];
            a[j] = VAR == 2 ? rowbase + (uint32_t)p.x : lds0 + (uint32_t)(((p.x >> 10) << 10) + lane * 16);
        }
        for (int r = 0; r < records * 32 / NR; ++r) {
            f32x4 x[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) rd128<0>(x[j], a[j]);
#pragma unroll
            for (int q = 0; q < NR; q += 4) {
                if (NR - q - 4 == 12) w4<12>(x[q], x[q + 1], x[q + 2], x[q + 3]);
                else if (NR - q - 4 == 8) w4<8>(x[q], x[q + 1], x[q + 2], x[q + 3]);
                else if (NR - q - 4 == 4) w4<4>(x[q], x[q + 1], x[q + 2], x[q + 3]);
                else w4<0>(x[q], x[q + 1], x[q + 2], x[q + 3]);
#pragma unroll
                for (int j = q; j < q + 4; ++j) fma_row(acc[j], 1.0f, x[j]);
            }
        }
    } else if constexpr (VAR == 4) {   // E: ds_read_b64, one row per wave, 16 per wait (32 rows per "record" and wave)
        uint32_t a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int2 p = reinterpret_cast<const int2 *>(smem + kPanelBytes)[((threadIdx.x >> 6) * 16 + j) % 1024];
            a[j] = lds0 + (uint32_t)p.x + lane * 8;
        }
        for (int r = 0; r < records * 2; ++r) {
            f32x2 x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) rd64<0>(x[j], a[j]);
            w4h<12>(x[0], x[1], x[2], x[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][0] = __builtin_elementwise_fma(f32x2{1.f, 1.f}, x[j], acc[j][0]);
            w4h<8>(x[4], x[5], x[6], x[7]);
#pragma unroll
            for (int j = 4; j < 8; ++j) acc[j][0] = __builtin_elementwise_fma(f32x2{1.f, 1.f}, x[j], acc[j][0]);
            w4h<4>(x[8], x[9], x[10], x[11]);
#pragma unroll
            for (int j = 8; j < 12; ++j) acc[j][0] = __builtin_elementwise_fma(f32x2{1.f, 1.f}, x[j], acc[j][0]);
            w4h<0>(x[12], x[13], x[14], x[15]);
#pragma unroll
            for (int j = 12; j < 16; ++j) acc[j][0] = __builtin_elementwise_fma(f32x2{1.f, 1.f}, x[j], acc[j][0]);
        }
    } else if constexpr (VAR == 5) {   // F: shipped step, pairs by ds_read_b64 (one pair per read)
        for (int r = 0; r < records; ++r) {
            f32x2 q0, q1, q2, q3, m0, m1, m2, m3;
            rd64<0>(q0, pa); rd64<8>(q1, pa); rd64<16>(q2, pa); rd64<24>(q3, pa);
#define STEPF(J, MORE)                                                        \
            {                                                                 \
                w4h<0>(q0, q1, q2, q3);                                       \
                f32x4 x0, x1, x2, x3;                                         \
                rd128<0>(x0, rowbase + (uint32_t)__float_as_int(q0.x));       \
                rd128<0>(x1, rowbase + (uint32_t)__float_as_int(q1.x));       \
                rd128<0>(x2, rowbase + (uint32_t)__float_as_int(q2.x));       \
                rd128<0>(x3, rowbase + (uint32_t)__float_as_int(q3.x));       \
                if (MORE) {                                                   \
                    rd64<((J) + 2) * 16>(m0, pa); rd64<((J) + 2) * 16 + 8>(m1, pa); \
                    rd64<((J) + 3) * 16>(m2, pa); rd64<((J) + 3) * 16 + 8>(m3, pa); \
                    w4<4>(x0, x1, x2, x3);                                    \
                } else {                                                      \
                    w4<0>(x0, x1, x2, x3);                                    \
                }                                                             \
                fma_row(acc[(J)], q0.y, x0);                                  \
                fma_row(acc[(J)], q1.y, x1);                                  \
                fma_row(acc[(J) + 1], q2.y, x2);                              \
                fma_row(acc[(J) + 1], q3.y, x3);                              \
                if (MORE) { q0 = m0; q1 = m1; q2 = m2; q3 = m3; }             \
            }
            STEPF(0, true) STEPF(2, true) STEPF(4, true) STEPF(6, true) STEPF(8, true) STEPF(10, true) STEPF(12, true) STEPF(14, false)
#undef STEPF
        }
    } else if constexpr (VAR == 6) {   // G: rows of step J+1 and pairs of step J+2 in flight behind the FMAs of step J
        for (int r = 0; r < records; ++r) {
            f32x4 p0, p1, n0, n1, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3;
            rd128<0>(p0, pa);
            rd128<16>(p1, pa);
            w2<0>(p0, p1);
            rd128<0>(xa0, rowbase + (uint32_t)__float_as_int(p0.x));
            rd128<0>(xa1, rowbase + (uint32_t)__float_as_int(p0.z));
            rd128<0>(xa2, rowbase + (uint32_t)__float_as_int(p1.x));
            rd128<0>(xa3, rowbase + (uint32_t)__float_as_int(p1.z));
            rd128<32>(n0, pa);
            rd128<48>(n1, pa);
#define STEPG(J, XA0, XA1, XA2, XA3, XB0, XB1, XB2, XB3, MORE, MORE2)          \
            {                                                                 \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA0), "+v"(XA1), "+v"(XA2), "+v"(XA3), "+v"(n0), "+v"(n1)); \
                const float wa = p0.y, wb = p0.w, wc = p1.y, wd = p1.w;       \
                if (MORE) {                                                   \
                    rd128<0>(XB0, rowbase + (uint32_t)__float_as_int(n0.x));  \
                    rd128<0>(XB1, rowbase + (uint32_t)__float_as_int(n0.z));  \
                    rd128<0>(XB2, rowbase + (uint32_t)__float_as_int(n1.x));  \
                    rd128<0>(XB3, rowbase + (uint32_t)__float_as_int(n1.z));  \
                    p0 = n0; p1 = n1;                                         \
                    if (MORE2) { rd128<((J) + 4) * 16>(n0, pa); rd128<((J) + 5) * 16>(n1, pa); } \
                }                                                             \
                __builtin_amdgcn_sched_barrier(0);                            \
                fma_row(acc[(J)], wa, XA0);                                   \
                fma_row(acc[(J)], wb, XA1);                                   \
                fma_row(acc[(J) + 1], wc, XA2);                               \
                fma_row(acc[(J) + 1], wd, XA3);                               \
                __builtin_amdgcn_sched_barrier(0);                            \
            }
            STEPG(0, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3, true, true)
            STEPG(2, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3, true, true)
            STEPG(4, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3, true, true)
            STEPG(6, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3, true, true)
            STEPG(8, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3, true, true)
            STEPG(10, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3, true, true)
            STEPG(12, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3, true, false)
            STEPG(14, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3, false, false)
#undef STEPG
        }
    }
    float4 o = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) { o.x += acc[j][0].x; o.y += acc[j][0].y; o.z += acc[j][1].x; o.w += acc[j][1].y; }
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = o;
}

template <int VAR, int THREADS>
void run(const char *name, const int2 *pairs, float4 *out, int ncu, double clk_ghz) {
    const int records = 4000;
    CHECK(hipFuncSetAttribute((const void *)lds_rate<VAR, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((lds_rate<VAR, THREADS>), dim3(ncu), dim3(THREADS), kSmem, 0, pairs, out, records);
    CHECK(hipEventRecord(e0));
    const int reps = 3;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((lds_rate<VAR, THREADS>), dim3(ncu), dim3(THREADS), kSmem, 0, pairs, out, records);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    // row bytes per record and wave: 32 row reads x 1 KB (E: 32 reads x 512 B)
    const double waves = THREADS / 64.0;
    const double row_bytes = (double)records * waves * 32 * (VAR == 4 ? 512 : 1024);
    const double pair_bytes = (VAR == 0 || VAR == 1 || VAR == 6) ? (double)records * waves * 16 * 1024 : (VAR == 5 ? (double)records * waves * 32 * 512 : 0);
    const double clks = ms * 1e-3 * clk_ghz * 1e9;
    printf("%-44s %4d thr  %.3f ms  rows %.1f B/clk/CU (%.1f TB/s chip)  rows+pairs %.1f B/clk/CU   clk/record %.0f\n", name, THREADS, ms,
           row_bytes / clks, row_bytes * ncu / (ms * 1e-3) / 1e12, (row_bytes + pair_bytes) / clks, clks / records);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("device %s  CUs %d  clock %.2f GHz\n", prop.name, ncu, ghz);
    std::vector<int2> hp(1024);
    srand(1);
    for (auto &p : hp) { p.x = (rand() % 128) * 512; float w = 0.5f; p.y = *reinterpret_cast<int *>(&w); }
    int2 *pairs; float4 *out;
    CHECK(hipMalloc(&pairs, hp.size() * sizeof(int2)));
    CHECK(hipMemcpy(pairs, hp.data(), hp.size() * sizeof(int2), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)ncu * 1024 * sizeof(float4)));
    run<0, 1024>("A shipped step (2 pair b128 + 4 rows / wait)", pairs, out, ncu, ghz);
    run<6, 1024>("G explicit pipeline (rows J+1 behind FMAs J)", pairs, out, ncu, ghz);
    run<2, 1024>("C rows only, 16 / wait, random rows", pairs, out, ncu, ghz);
    run<3, 1024>("D rows only, 16 / wait, wave-linear 1 KB", pairs, out, ncu, ghz);
    run<4, 1024>("E ds_read_b64 row per wave, 16 / wait", pairs, out, ncu, ghz);
    run<5, 1024>("F shipped step, pairs by 4 x b64", pairs, out, ncu, ghz);
    run<0, 512>("A shipped step", pairs, out, ncu, ghz);
    run<6, 512>("G explicit pipeline", pairs, out, ncu, ghz);
    run<1, 512>("B 8 rows / wait", pairs, out, ncu, ghz);
    run<2, 512>("C rows only random", pairs, out, ncu, ghz);
    run<3, 512>("D rows only linear", pairs, out, ncu, ghz);
    run<4, 512>("E b64 row per wave", pairs, out, ncu, ghz);
    run<2, 256>("C rows only random", pairs, out, ncu, ghz);
    run<4, 256>("E b64 row per wave", pairs, out, ncu, ghz);
    return 0;
}
