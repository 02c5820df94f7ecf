// Register-only MFMA loop: what the matrix pipes of this MI355X sustain with no memory traffic at all.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak ; run: tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int NACC>
__global__ __launch_bounds__(512) void k16(const unsigned* seed, float* out, int iters) {
    unsigned s = seed[threadIdx.x & 63] * 2654435761u + threadIdx.x;
    bf8 a[8], b[5];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(s >> 16) & 255) / 128.f - 1.f); }
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(s >> 16) & 255) / 128.f - 1.f); }
    f4 acc[8][5];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    f4 t{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) t += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

__global__ __launch_bounds__(512) void k32(const unsigned* seed, float* out, int iters) {
    unsigned s = seed[threadIdx.x & 63] * 2654435761u + threadIdx.x;
    bf8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(s >> 16) & 255) / 128.f - 1.f); }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(s >> 16) & 255) / 128.f - 1.f); }
    f16v acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float t = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
    unsigned hs[64]; for (int i = 0; i < 64; ++i) hs[i] = 12345u + 977u * i;
    unsigned* ds; float* out;
    hipMalloc(&ds, sizeof hs); hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
    hipMalloc(&out, 4096 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        for (int threads : {256, 512}) {
            for (int which = 0; which < 2; ++which) {
                const int iters = 4000;
                float best = 1e30f, last = 0;
                for (int rep = 0; rep < 6; ++rep) {
                    hipEventRecord(e0);
                    if (which == 0) hipLaunchKernelGGL(k16<40>, dim3(blocks), dim3(threads), 0, 0, ds, out, iters);
                    else hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, ds, out, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms; last = ms;
                }
                const double waves = (double)blocks * threads / 64;
                const double flops = waves * iters * (which == 0 ? 40.0 * 16 * 16 * 32 * 2 : 8.0 * 32 * 32 * 16 * 2);
                printf("%s blocks=%d threads=%d  best %.3f ms  %.1f TF   (last %.3f ms %.1f TF)\n", which ? "32x32x16" : "16x16x32",
                       blocks, threads, best, flops / best / 1e9, last, flops / last / 1e9);
            }
        }
    }
    // sustained: 200 back-to-back launches (clock behaviour under load)
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k16<40>, dim3(256), dim3(512), 0, 0, ds, out, 4000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("sustained 16x16x32 256x512 x200: %.1f ms  %.1f TF\n", ms, 200.0 * 256 * 8 * 4000 * 40.0 * 16 * 16 * 32 * 2 / ms / 1e9);
    return 0;
}
