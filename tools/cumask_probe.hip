// Dev tool: which XCDs / CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Each workgroup records its
// XCC id and HW id; the host prints, per mask, the number of workgroups seen on each XCD and the distinct (se, sh, cu) count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void probe(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);   // keep the CU busy so the grid spreads
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
int main() {
    const int NWG = 1024;
    unsigned* d; CK(hipMalloc(&d, NWG * 8));
    std::vector<unsigned> h(NWG * 2);
    struct M { const char* name; uint32_t w[8]; };
    std::vector<M> masks;
    { M m{"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}}; masks.push_back(m); }
    { M m{"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
    { M m{"bits 0..63", {~0u, ~0u, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
    { M m{"bits = 0 mod 8", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}}; masks.push_back(m); }
    { M m{"bits = 0,1 mod 8", {0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u}}; masks.push_back(m); }
    { M m{"bits 128..255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}}; masks.push_back(m); }
    for (auto& m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (e != hipSuccess) { printf("mask %-18s: create failed: %s\n", m.name, hipGetErrorString(e)); continue; }
        CK(hipMemsetAsync(d, 0xff, NWG * 8, s));
        // 64 KB of LDS per workgroup would limit co-residency; here 1024 threads per workgroup = 2 workgroups per CU at most
        hipLaunchKernelGGL(probe, dim3(NWG), dim3(1024), 0, s, d, 200000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, NWG * 8, hipMemcpyDeviceToHost));
        int per_xcc[16] = {0};
        std::set<unsigned> cus;
        for (int i = 0; i < NWG; ++i) {
            const unsigned x = h[2 * i] & 0xf, hw = h[2 * i + 1];
            per_xcc[x]++;
            cus.insert((x << 16) | (hw & 0xff00));   // cu_id[11:8], sh_id[12], se_id[15:13]
        }
        printf("mask %-18s: workgroups per XCD:", m.name);
        for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
        printf("   distinct (xcd, se, sh, cu): %zu\n", cus.size());
        CK(hipStreamDestroy(s));
    }
    return 0;
}
