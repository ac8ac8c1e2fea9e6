// cumask_probe.hip -- which CUs does bit i of a hipExtStreamCreateWithCUMask mask select on an MI355X (8 XCDs x 32 CUs)?
// Every workgroup records {XCC_ID, HW_ID}; the host prints, per mask, how many distinct CUs of each XCD ran workgroups and
// whether workgroup i still lands on XCD i % 8 (what the XCD-sliced gather plan relies on).
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/cumask_probe.hip -o tools/micro/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>

__global__ void who(uint32_t *out, int spin) {
    if (threadIdx.x == 0) {
        uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));       // HW_REG_XCC_ID [3:0]
        uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(20);
}

static void run(const char *name, int lo, int hi) {
    uint32_t mask[8] = {0};
    for (int i = lo; i < hi; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int nb = 8192;
    uint32_t *d; hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(who, dim3(nb), dim3(256), 0, s, d, 200);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(nb * 2); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus[16]; int onmod = 0;
    for (int b = 0; b < nb; ++b) {
        uint32_t x = h[2 * b] & 15, hw = h[2 * b + 1];
        cus[x].insert((hw >> 8) & 0xff);          // cu_id[11:8] sh_id[12] se_id[15:13]
        onmod += (int)(x == (uint32_t)(b % 8));
    }
    printf("%-28s bits [%3d,%3d): CUs per XCD", name, lo, hi);
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); tot += cus[x].size(); }
    printf("  total %3d   workgroup i on XCD i%%8: %.1f%%\n", tot, 100.0 * onmod / nb);
    hipFree(d); hipStreamDestroy(s);
}

int main() {
    run("all", 0, 256); run("first 64 bits", 0, 64); run("first 128 bits", 0, 128); run("bits 192..256", 192, 256);
    run("first 8 bits", 0, 8); run("bits 8..16", 8, 16); run("first 32 bits", 0, 32); run("first 192 bits", 0, 192);
    return 0;
}
