// Dev microbenchmark: sustained MFMA rate with CHANGING random operands (realistic switching activity), bf16 16x16x32 vs
// fp8 (e4m3) 16x16x128 f8f6f4, plus the shader clock (s_memtime / wall clock) while it runs.  The register-only loop of
// mfma_peak.hip reuses one operand pair, which under-states power; the model's GEMMs run power-limited.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int KIND>   // 0: bf16 16x16x32, 1: fp8 16x16x128
__global__ __launch_bounds__(512) void k(const unsigned* seed, float* out, unsigned* clk, int iters) {
    // 8 A fragments x 5 B fragments per "K-step", 4 different operand sets rotated per iteration
    unsigned s = seed[threadIdx.x & 63] * 2654435761u + threadIdx.x * 97u + blockIdx.x;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    f4 acc[8][5];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = f4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if constexpr (KIND == 0) {
        bf8 a[2][8], b[2][5];
        for (int v = 0; v < 2; ++v) {
            for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[v][i][e] = (__bf16)(((int)(rnd() >> 16) & 1023) / 512.f - 1.f);
            for (int i = 0; i < 5; ++i) for (int e = 0; e < 8; ++e) b[v][i][e] = (__bf16)(((int)(rnd() >> 16) & 1023) / 512.f - 1.f);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[v][i], b[v][j], acc[i][j], 0, 0, 0);
        }
    } else {
        i8v a[2][4], b[2][3];           // fewer distinct fragments (32 B each): 4 x 3 tiles reused over the 8 x 5 accumulators
        for (int v = 0; v < 2; ++v) {
            for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[v][i][e] = (int)(rnd() & 0x77777777u);   // e4m3 without the top exponent bit: finite, moderate
            for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) b[v][i][e] = (int)(rnd() & 0x77777777u);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 5; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[v][i & 3], b[v][j % 3], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    f4 t{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) t += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = (unsigned)(t1 - t0); clk[1] = (unsigned)(w1 - w0); }
}

int main() {
    unsigned hs[64]; for (int i = 0; i < 64; ++i) hs[i] = 12345u + 977u * i;
    unsigned* ds; float* out; unsigned* clk;
    hipMalloc(&ds, sizeof hs); hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind) {
        const int iters = 4000, launches = 100;          // ~0.5 s sustained
        hipEventRecord(e0);
        for (int l = 0; l < launches; ++l) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, ds, out, clk, iters);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, ds, out, clk, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned hc[2]; hipMemcpy(hc, clk, 8, hipMemcpyDeviceToHost);
        const double flops = (double)launches * 256 * 8 * iters * 80.0 * 16 * 16 * (kind == 0 ? 32 : 128) * 2;
        printf("%s random operands, %d launches back to back: %.1f ms  %.0f TFLOP/s; shader clock in the last launch %.0f MHz\n",
               kind == 0 ? "bf16 16x16x32 " : "fp8  16x16x128", launches, ms, flops / ms / 1e9, hc[0] / (hc[1] * 0.01));
    }
    return 0;
}
