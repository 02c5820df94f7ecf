// Dev microbenchmark: VALU work interleaved IN THE SAME WAVE between MFMAs: how much fits in the MFMA shadow (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;

template <int NF, int KIND, int MF>   // NF VALU ops after every MFMA; KIND 0 = v_fma, 1 = v_exp (+mul), 2 = v_pk_fma; MF 0 = 32x32x16, 1 = 16x16x32
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x + j) & 3); b[j] = (__bf16)(float)((threadIdx.x * 3 + j) & 3); }
    f16v acc[4]; f4v acc4[4];
    for (int i = 0; i < 4; ++i) { for (int e = 0; e < 16; ++e) acc[i][e] = 0.f; acc4[i] = f4v{0, 0, 0, 0}; }
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const float c0 = 1.0001f, c1 = 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MF == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int q = (i * NF + j) & 31;
                if (KIND == 0) x[q] = fmaf(x[q], c0, c1);
                else if (KIND == 1) x[q] = __builtin_amdgcn_exp2f(x[q]);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (NF > 0) __builtin_amdgcn_sched_group_barrier(0x002, NF, 0);
        }
    }
    float t = 0;
    for (int i = 0; i < 4; ++i) { for (int e = 0; e < 16; ++e) t += acc[i][e]; t += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3]; }
    for (int i = 0; i < 32; ++i) t += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NF, int KIND, int MF> void run(float* out, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL((k<NF, KIND, MF>), dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NF, KIND, MF>), dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s waves/SIMD=%d: %7.1f ns per 4 MFMAs per wave\n", what, threads / 256, ms * 1e6 / iters);
    }
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 0, 0>(out, "32x32x16 alone");
    run<4, 0, 0>(out, "32x32x16 + 4 v_fma each");
    run<6, 0, 0>(out, "32x32x16 + 6 v_fma each");
    run<8, 0, 0>(out, "32x32x16 + 8 v_fma each");
    run<12, 0, 0>(out, "32x32x16 + 12 v_fma each");
    run<2, 1, 0>(out, "32x32x16 + 2 v_exp each");
    run<4, 1, 0>(out, "32x32x16 + 4 v_exp each");
    run<0, 0, 1>(out, "16x16x32 alone");
    run<2, 0, 1>(out, "16x16x32 + 2 v_fma each");
    run<3, 0, 1>(out, "16x16x32 + 3 v_fma each");
    run<4, 0, 1>(out, "16x16x32 + 4 v_fma each");
    return 0;
}
