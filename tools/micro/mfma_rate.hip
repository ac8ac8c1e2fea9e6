// Issue rate of v_mfma_f32_32x32x16_bf16 under the dependency patterns of the dense3 loop (r03, prepared for r04).
// The harness run of tools/experiments/dense3 found its MFMA loop at 46 % of the matrix rate with nothing else in the
// loop (profiles/r03_dense3_bench.txt): every number fits "this MFMA takes 64 cycles here" -- this binary separates
// the candidates in one two-second run.  Per wave, N MFMAs in a loop, timed with s_memtime (ticks = shader cycles):
//   mode 0: four accumulators round robin        mode 1: two accumulators alternating (the dense3 half step)
//   mode 2: one accumulator, a dependent chain   mode 3: mode 1 with one ds_read_b128 per two MFMAs feeding B
//   mode 4: mode 1, B operands rewritten by a v_mov between uses (operand freshly written by the VALU)
// each with one and with two waves per SIMD (256 / 512 workgroups of 256 threads).  Prints ticks per MFMA and wave.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ f32x16 mma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate_kernel(int iters, unsigned long long *ticks, float *sink) {
    __shared__ u32x4 lds[256 * 4];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256 * 4; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    u32x4 a = {0x3f803f80u + (uint32_t)lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 b0 = a, b1 = a;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = mma(a, b0, acc[0]); acc[1] = mma(a, b1, acc[1]); acc[2] = mma(a, b0, acc[2]); acc[3] = mma(a, b1, acc[3]);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc[0] = mma(a, b0, acc[0]); acc[1] = mma(a, b1, acc[1]); }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[0] = mma(a, b0, acc[0]);
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                b0 = lds[(threadIdx.x + 64 * u + 13 * it) & 1023];        // (varies with the trip: stays in the loop)
                acc[0] = mma(a, b0, acc[0]); acc[1] = mma(a, b1, acc[1]);
                b1 = b0;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[0] = mma(a, b0, acc[0]); acc[1] = mma(a, b1, acc[1]);
                b0[0] += 1u; b1[1] += 1u;                   // operands freshly written by the VALU
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    if (lane == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;                        // keeps the accumulators alive
}

template <int MODE>
static void run(const char *what, int nwg, int iters, unsigned long long *dt, float *sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(nwg), dim3(256), 0, nullptr, iters, dt, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(nwg), dim3(256), 0, nullptr, iters, dt, sink);
    CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t((size_t)nwg * 4);
    CHECK(hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto x : t) m += (double)x;
    m /= (double)t.size();
    const double n = 16.0 * iters;
    printf("%-52s %4d workgroups: %7.1f ticks per MFMA and wave, kernel %8.1f us (%.1f ns per MFMA and wave)\n", what, nwg, m / n, 1e3 * ms,
           1e6 * ms / n);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; %d x 16 MFMAs (v_mfma_f32_32x32x16_bf16: 8 passes = 32 cycles at the matrix rate) per wave\n", prop.gcnArchName,
           prop.multiProcessorCount, iters);
    unsigned long long *dt; float *sink;
    CHECK(hipMalloc(&dt, 8 * 4 * 1024)); CHECK(hipMalloc(&sink, 4));
    for (int nwg : {256, 512}) {
        run<0>("four accumulators round robin", nwg, iters, dt, sink);
        run<1>("two accumulators alternating (dense3 half step)", nwg, iters, dt, sink);
        run<2>("one accumulator (dependent chain)", nwg, iters, dt, sink);
        run<3>("two accumulators, one ds_read_b128 per two MFMAs", nwg, iters, dt, sink);
        run<4>("two accumulators, B rewritten by the VALU", nwg, iters, dt, sink);
    }
    return 0;
}
