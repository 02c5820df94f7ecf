// Dev microbenchmark: does VALU work of one wave run under the MFMAs of the other wave on the same SIMD (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run VALU FMAs; bit 2: waves 4-7 run v_exp_f32; bit 3: every wave interleaves
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x + j) & 3); b[j] = (__bf16)(float)((threadIdx.x * 3 + j) & 3); }
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const float c0 = 1.0001f, c1 = 1e-4f;
    if ((mode & 16) && wave >= 4) __builtin_amdgcn_s_setprio(3);
    if ((mode & 32) && wave < 4) __builtin_amdgcn_s_setprio(3);
    if (mode & 8) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[i * 4 + j] = fmaf(x[i * 4 + j], c0, c1);
            }
        }
    } else if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], c0, c1);
            }
        if (mode & 4)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i] * 1e-3f);
            }
    }
    float t = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
    for (int i = 0; i < 16; ++i) t += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[] = {"", "MFMA waves only (4 x 32x32x16 per iter)", "VALU waves only (32 v_fma per iter)", "MFMA waves + VALU waves", "exp waves only (8 v_exp + 8 v_mul per iter)",
                           "MFMA waves + exp waves", "", "", "every wave: 4 MFMA interleaved with 16 v_fma"};
    for (int mode : {1, 2, 3, 19, 35, 4, 5, 8}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %2d %-48s %8.3f ms  (%.1f ns per iteration)\n", mode, mode == 19 ? "MFMA waves + VALU waves at prio 3" : mode == 35 ? "MFMA waves at prio 3 + VALU waves" : names[mode], ms, ms * 1e6 / iters);
    }
    return 0;
}
