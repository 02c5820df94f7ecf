// Dev microbenchmark: issue rate of the block-scaled fp8 MFMAs (register-only loop), cycles per instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma8_rate.hip -o tools/mfma8_rate && tools/mfma8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(const int* seed, float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = seed[(l * 8 + j) & 1023]; b[j] = seed[(l * 8 + j + 77) & 1023]; }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 acc4[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f;
    const int one = 0x7f7f7f7f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0, 0, 0, one, 0, one);
            else if constexpr (KIND == 1) acc4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc4[i], 0, 0, 0, one, 0, one);
            else if constexpr (KIND == 2) {
                bf16x8 x = __builtin_bit_cast(bf16x8, (__attribute__((ext_vector_type(4))) int){a[0], a[1], a[2], a[3]});
                bf16x8 y = __builtin_bit_cast(bf16x8, (__attribute__((ext_vector_type(4))) int){b[0], b[1], b[2], b[3]});
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
            } else {
                // fp8 non-scaled 32x32x16 (bf16 rate)
                long x = ((long)a[1] << 32) | (unsigned)a[0], y = ((long)b[1] << 32) | (unsigned)b[0];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(x, y, acc[i], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) { for (int e = 0; e < 16; ++e) s += acc[i][e]; for (int e = 0; e < 4; ++e) s += acc4[i][e]; }
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int threads, double flop_per_mfma) {
    const int iters = 2000, grid = 256;
    int* seed; float* out; unsigned long long* cyc;
    hipMalloc(&seed, 4096); hipMalloc(&out, grid * threads * 4); hipMalloc(&cyc, grid * 8);
    std::vector<int> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x38383838 ^ (i * 2654435761u & 0x07070707);
    hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, seed, out, cyc, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, seed, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 256;
    const double cyc_per = (double)c[0] / (iters * 8.0) / waves_per_simd;      // per SIMD: waves share it
    const double total_mfma = 5.0 * grid * (threads / 64) * iters * 8.0;
    printf("%-28s %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD (wave view %.1f); %.0f TFLOP/s, implied clock %.0f MHz\n", name, waves_per_simd,
           cyc_per, (double)c[0] / (iters * 8.0), total_mfma * flop_per_mfma / (ms * 1e-3) / 1e12, (double)c[0] / (ms / 5 * 1e3));
}

int main() {
    for (int t : {256, 512}) {
        run<0>("scale 32x32x64 f8f6f4 (fp8)", t, 2.0 * 32 * 32 * 64);
        run<1>("scale 16x16x128 f8f6f4 (fp8)", t, 2.0 * 16 * 16 * 128);
        run<2>("32x32x16 bf16", t, 2.0 * 32 * 32 * 16);
        run<3>("32x32x16 fp8 (non-scaled)", t, 2.0 * 32 * 32 * 16);
    }
    return 0;
}
