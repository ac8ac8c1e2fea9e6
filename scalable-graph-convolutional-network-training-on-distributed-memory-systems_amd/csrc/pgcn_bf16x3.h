// pgcn_bf16x3.h -- what the kernels on the bf16 matrix cores share (pgcn_spmm_dense3.hip, pgcn_gat_blocks.hip): the exact three-plane
// split of an fp32 number, the MFMA, LDS reads the compiler does not see, and the kernel that splits the panels of the dense operand
// into the LDS image of a block kernel.  Device code only; everything has internal linkage (each translation unit its own copy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pgcn_bf16x3 {

constexpr int kT = 128;                  // columns of a block = rows of a panel = features per feature block
constexpr int kSplitThreads = 256;
constexpr int kQBytes = 3 * 4 * kT * 16; // one quarter image: 3 planes x 4 k groups x 128 columns x 16 B = 24 KB
constexpr int kImgBytes = 4 * kQBytes;   // one panel x feature block: 96 KB

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {     // {bf16(x) in bits 0-15, bf16(y) in bits 16-31}, RNE
    const f32x2 v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_as_f32(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_as_f32(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// x, y -> the three bf16 planes of both, packed {x in bits 0-15, y in bits 16-31}:  x = x1 + x2 + x3, x1 = bf16(x), x2 = bf16(x - x1),
// x3 = x - x1 - x2 (both remainders exact fp32 subtractions, the last one a bf16 number)
__device__ __forceinline__ void split_pair(float x, float y, uint32_t &u1, uint32_t &u2, uint32_t &u3) {
    u1 = pack_bf16(x, y);
    const float rx = x - lo_as_f32(u1), ry = y - hi_as_f32(u1);          // exact
    u2 = pack_bf16(rx, ry);
    u3 = pack_bf16(rx - lo_as_f32(u2), ry - hi_as_f32(u2));              // exact, and a bf16 number
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the six partial products that matter of (a1 + a2 + a3)(b1 + b2 + b3), smallest first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
#define PGCN_BF16X3_PRODUCTS constexpr int kPA[6] = {2, 0, 1, 1, 0, 0}, kPB[6] = {0, 2, 1, 0, 1, 0}

// LDS reads the compiler does not see as memory operations (a ds_read it can see makes it wait for ALL outstanding asynchronous
// global -> LDS copies); the matching waits take the results as read-write operands
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4 &v, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128(f32x4 &v, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- the panel split ------------------------------------------------------------------------------------------
// The panels of the dense operand that blocks refer to, split ONCE per product into the exact LDS image of the block kernels:
//     image[panel][feature block] = [quarter q (32 k)][plane p][k group kg (8 k)][column n (128)] x 16 B
// (8 bf16: k = 32 q + 8 kg + j), i.e. a lane's B operand of v_mfma_f32_32x32x16_bf16 is one 16-byte slot and the 32 lanes of a half
// wave read 512 contiguous bytes.  grid (panels in the list, feature blocks of 128); thread t: column n = t & 127, k groups
// 8 (t >> 7) .. + 8.  panel_list holds the FIRST ROW of every panel (any row: a grid aligned to the vertex order's bands).
static __global__ __launch_bounds__(kSplitThreads) void spmm_split_panels_kernel(const int32_t *__restrict__ panel_list, const float *__restrict__ B,
                                                                            int64_t ldb, int64_t ncols, int32_t f, u32x4 *__restrict__ image) {
    const int64_t r0 = (int64_t)panel_list[blockIdx.x];
    const int fcol0 = blockIdx.y * kT;
    const int n = threadIdx.x & (kT - 1);
    const bool n_ok = fcol0 + n < f;
    const float *col = B + fcol0 + (n_ok ? n : 0);
    u32x4 *img = image + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (kImgBytes / 16);
    const int kg0 = (threadIdx.x >> 7) * 8;
#pragma unroll 2
    for (int kgi = 0; kgi < 8; ++kgi) {
        const int kga = kg0 + kgi;                         // k group of the panel: rows 8 kga .. 8 kga + 7
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = r0 + 8 * kga + j;
            x[j] = (n_ok && r < ncols) ? col[r * ldb] : 0.f;
        }
        u32x4 p1, p2, p3;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t u1, u2, u3;
            split_pair(x[2 * d], x[2 * d + 1], u1, u2, u3);
            p1[d] = u1; p2[d] = u2; p3[d] = u3;
        }
        const int q = kga >> 2, kg = kga & 3;
        u32x4 *dst = img + ((q * 3) * 4 + kg) * kT + n;
        dst[0] = p1;
        dst[4 * kT] = p2;
        dst[8 * kT] = p3;
    }
}

}  // namespace pgcn_bf16x3
