// Microbenchmark (gfx950): can one wave's MFMAs and ANOTHER wave's vector ALU work on the same SIMD overlap, or only MFMA + VALU
// interleaved in ONE wave's instruction stream?  One workgroup per CU, 8 waves (2 per SIMD: waves w and w + 4 share SIMD w % 4).
//   mode 0: waves 0-3 run MFMAs (32x32x16 f16), waves 4-7 idle          -> T_m
//   mode 1: waves 0-3 idle, waves 4-7 run VALU (v_fma_f32)               -> T_v
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (partners in different phases) -> max(T_m, T_v) if they overlap, T_m + T_v if not
//   mode 3: all 8 waves run an interleaved stream, half the iterations each (1 MFMA : VPM VALU)
//   mode 4: mode 2 with v_exp_f32 instead of v_fma_f32
//   mode 5: mode 3 on waves 0-3 only (one wave per SIMD, all iterations)
//   mode 6: VALU only on all 8 waves, half the iterations each (what two waves per SIMD issue per cycle)
// build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_valu_overlap tools/experiments/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VPM>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool mf = (mode == 0 || mode == 2 || mode == 4) ? wave < 4 : (mode == 3);
    const bool va = (mode == 1 || mode == 2 || mode == 4) ? wave >= 4 : (mode == 3 || mode == 6);
    if (mode == 5 && wave >= 4) return;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[4] = {};
    float v[VPM];
    for (int i = 0; i < VPM; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float c0 = 1.0001f, c1 = 0.5f;
    if (mode == 3 || mode == 5) {
        for (int it = 0; it < (mode == 3 ? iters / 2 : iters); ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < VPM; ++i) v[i] = __builtin_fmaf(v[i], c0, c1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
        }
    } else if (mf) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        }
    } else if (va) {
        if (mode == 4) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int i = 0; i < VPM; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) ;
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int i = 0; i < VPM; ++i) v[i] = __builtin_fmaf(v[i], c0, c1);
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int i = 0; i < 16; ++i) s += acc[m][i];
    for (int i = 0; i < VPM; ++i) s += v[i];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (mode == 5) { if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0; return; }
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;      // shader-clock cycles of the whole workgroup (s_memtime)
}

template <int VPM>
void run(const char* name, int mode, int iters) {
    float* out; hipMalloc(&out, 4096);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best_us = 1e30, best_cyc = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<VPM>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i];
        if (rep) { best_us = ms * 1000 < best_us ? ms * 1000 : best_us; best_cyc = c / 256 < best_cyc ? c / 256 : best_cyc; }
    }
    printf("VALU per MFMA %2d  mode %d %-44s %8.1f us  %9.0f cycles per workgroup (%.1f per MFMA slot)\n", VPM, mode, name, best_us, best_cyc, best_cyc / (iters * 4.0));
    hipFree(out); hipFree(cyc);
}
template <int VPM> void all(int iters) {
    run<VPM>("MFMA waves alone", 0, iters);
    run<VPM>("VALU waves alone", 1, iters);
    run<VPM>("MFMA waves + VALU waves (SIMD partners)", 2, iters);
    run<VPM>("one stream, interleaved, all 8 waves", 3, iters);
    run<VPM>("MFMA waves + v_exp waves (SIMD partners)", 4, iters);
    run<VPM>("one stream, interleaved, waves 0-3 only", 5, iters);
    run<VPM>("VALU on all 8 waves (2 per SIMD), half each", 6, iters / 2);
}
int main() {
    const int iters = 20000;
    all<2>(iters); all<4>(iters); all<6>(iters); all<7>(iters); all<8>(iters);
    return 0;
}
