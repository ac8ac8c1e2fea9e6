// Micro-benchmark: how fast can a CU gather 512-B feature rows out of an LDS-staged panel?
// Models the planned core kernel: WG stages TC rows x 128 floats into LDS, then every 32-lane
// group does E (col,val) FMAs per "row", cols random in [0,TC).  Prints aggregate TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int TC, int THREADS, bool META_LDS>
__global__ __launch_bounds__(THREADS) void lds_gather(const float4* __restrict__ H, const int2* __restrict__ meta,
                                                      float4* __restrict__ out, int panels, int edges_per_group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* panel = reinterpret_cast<float4*>(smem);                 // TC x 32 float4
    float2* mpark = reinterpret_cast<float2*>(smem + TC * 512);      // per wave 64 x 8 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 31, gbase = lane & 32;
    const int ngroups = THREADS / 32;
    const int group = threadIdx.x >> 5;
    float4 acc = make_float4(0, 0, 0, 0);
    const int2* mp = meta + ((size_t)blockIdx.x * ngroups + group) * edges_per_group;
    for (int p = 0; p < panels; ++p) {
        __syncthreads();
        const float4* src = H + (size_t)((blockIdx.x + p) % 64) * TC * 32;
        for (int i = threadIdx.x; i < TC * 32; i += THREADS) panel[i] = src[i];
        __syncthreads();
        int2 nx = mp[sub];
        for (int e = 0; e < edges_per_group; e += 32) {
            int2 cur = nx;
            int en = e + 32 + sub; en = en < edges_per_group ? en : edges_per_group - 1;
            nx = mp[en];
            if (META_LDS) {
                mpark[wave * 64 + lane] = make_float2(__int_as_float(cur.x), __int_as_float(cur.y));
#pragma unroll
                for (int k = 0; k < 32; k += 8) {
                    float4 m[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) m[u] = *reinterpret_cast<const float4*>(&mpark[wave * 64 + gbase + k + 2 * u]);
                    float4 x[8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        x[2 * u] = panel[(__float_as_int(m[u].x) << 5) + sub];
                        x[2 * u + 1] = panel[(__float_as_int(m[u].z) << 5) + sub];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc.x += m[u].y * x[2 * u].x; acc.y += m[u].y * x[2 * u].y; acc.z += m[u].y * x[2 * u].z; acc.w += m[u].y * x[2 * u].w;
                        acc.x += m[u].w * x[2 * u + 1].x; acc.y += m[u].w * x[2 * u + 1].y; acc.z += m[u].w * x[2 * u + 1].z; acc.w += m[u].w * x[2 * u + 1].w;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 32; k += 8) {
                    int c[8]; float w[8]; float4 x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { c[u] = __shfl(cur.x, gbase + k + u); w[u] = __int_as_float(__shfl(cur.y, gbase + k + u)); }
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = panel[(c[u] << 5) + sub];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { acc.x += w[u] * x[u].x; acc.y += w[u] * x[u].y; acc.z += w[u] * x[u].z; acc.w += w[u] * x[u].w; }
                }
            }
        }
    }
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <int TC, int THREADS, bool META_LDS>
void run(const char* name, int grid, int panels, int epg) {
    const int ngroups = THREADS / 32;
    float4* H; int2* meta; float4* out;
    CHECK(hipMalloc(&H, (size_t)64 * TC * 512));
    CHECK(hipMemset(H, 0, (size_t)64 * TC * 512));
    std::vector<int2> hm((size_t)grid * ngroups * epg);
    for (auto& m : hm) { m.x = rand() % TC; float f = 1.0f; m.y = *reinterpret_cast<int*>(&f); }
    CHECK(hipMalloc(&meta, hm.size() * sizeof(int2)));
    CHECK(hipMemcpy(meta, hm.data(), hm.size() * sizeof(int2), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)grid * THREADS * sizeof(float4)));
    size_t smem = TC * 512 + (THREADS / 64) * 512;
    CHECK(hipFuncSetAttribute((const void*)lds_gather<TC, THREADS, META_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((lds_gather<TC, THREADS, META_LDS>), dim3(grid), dim3(THREADS), smem, 0, H, meta, out, panels, epg);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((lds_gather<TC, THREADS, META_LDS>), dim3(grid), dim3(THREADS), smem, 0, H, meta, out, panels, epg);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double edges = (double)grid * ngroups * epg * panels;
    printf("%-34s grid %5d panels %3d edges/group/panel %5d : %.3f ms  LDS gather %.1f TB/s  staging %.2f TB/s\n", name, grid, panels, epg, ms,
           edges * 512 / ms / 1e9, (double)grid * panels * TC * 512 / ms / 1e9);
    CHECK(hipFree(H)); CHECK(hipFree(meta)); CHECK(hipFree(out));
}

int main() {
    // work per (WG, panel): ngroups * epg edges.  TC=128 (64 KB): 2 WGs/CU.  TC=64: 4 WGs/CU.
    run<128, 512, false>("TC128 T512 bpermute", 2048, 8, 256);
    run<128, 512, true>("TC128 T512 ldsmeta", 2048, 8, 256);
    run<128, 512, true>("TC128 T512 ldsmeta sparse", 2048, 8, 64);
    run<128, 256, true>("TC128 T256 ldsmeta", 2048, 8, 256);
    run<64, 256, true>("TC64 T256 ldsmeta", 4096, 8, 128);
    run<64, 256, false>("TC64 T256 bpermute", 4096, 8, 128);
    run<128, 1024, true>("TC128 T1024 ldsmeta", 1024, 8, 256);
    return 0;
}
