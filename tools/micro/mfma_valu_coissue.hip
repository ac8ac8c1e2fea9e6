// r06 probe: how many independent VALU instructions hide under one v_mfma_f32_32x32x16_bf16 (gfx950), with one and with two waves
// per SIMD?  Each wave runs ITER x [1 MFMA (alternating two accumulators) + N independent v_fma_f32 / v_exp_f32]; prints shader cycles
// per MFMA.   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coissue tools/micro/mfma_valu_coissue.hip && ./mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int N, int KIND>
__global__ __launch_bounds__(512) void k(float *out, long long *ticks, int iters) {
    f32x16 a0 = {}, a1 = {};
    u32x4 x = {threadIdx.x, 1, 2, 3}, y = {4, 5, 6, threadIdx.x};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), a1, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), a0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i & 15]) : "v"(v[(i + 5) & 15]));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 15]));
                else asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(v[(i + 5) & 15]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + a0[i] + a1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N, int KIND>
void run(int threads, const char *what) {
    float *out; long long *ticks;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&ticks, 1024 * 8 * 8);
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k<N, KIND>), dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, KIND>), dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    printf("%-6s N=%2d waves/SIMD=%d : %7.1f memtime ticks per MFMA (wave 0), %.3f ms -> %.1f ns per MFMA per wave\n", what, N, threads / 256,
           (double)h[0] / (iters * 8.0), ms, 1e6 * ms / (iters * 8.0));
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int threads : {256, 512}) {
        run<0, 0>(threads, "fma"); run<4, 0>(threads, "fma"); run<8, 0>(threads, "fma"); run<12, 0>(threads, "fma"); run<16, 0>(threads, "fma");
        run<24, 0>(threads, "fma"); run<8, 1>(threads, "exp"); run<8, 2>(threads, "and"); run<16, 2>(threads, "and");
    }
    return 0;
}
