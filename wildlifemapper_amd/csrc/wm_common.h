// Shared device helpers for the gfx950 kernels.  CDNA4 only: wave64, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define WM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// 16-bit operand traits: storage is always 2 bytes; only the MFMA opcode and
// the conversions differ between bf16 and fp16.
struct BF16 {
    using elem = __bf16;
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static __device__ __forceinline__ elem from_f32(float x) { return (__bf16)x; }
    static __device__ __forceinline__ float to_f32(elem x) { return (float)x; }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

struct FP16 {
    using elem = _Float16;
    using vec8 = f16x8;
    using vec4 = f16x4;
    static __device__ __forceinline__ elem from_f32(float x) {
        // saturate instead of producing inf: fp16 max is 65504
        x = fminf(fmaxf(x, -65504.f), 65504.f);
        return (_Float16)x;
    }
    static __device__ __forceinline__ float to_f32(elem x) { return (float)x; }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU of common.py:26 (nn.GELU default).  Two forms:
//   gelu_erf      - libm erff, used where the result stays fp32 (gemm32);
//   gelu_erf_fast - erf as a clamped rational x*P(x^2)/Q(x^2) (degree 6/4 in x^2, the form used by Eigen/XLA's float
//                   erf; max abs error 4.2e-7 against math.erf over [-6,6], checked on the host): 13 FMAs + one
//                   v_rcp_f32, all but the rcp packable, against two transcendentals and a branchy tail for erff.
//                   Used in the 16-bit GEMM epilogue, whose output is rounded to 2^-9 / 2^-12 relative anyway.
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
    float z = x * 0.70710678118654752440f;
    z = fminf(fmaxf(z, -4.0f), 4.0f);
    const float z2 = z * z;
    float p = -2.72614225801306e-10f;
    p = p * z2 + 2.77068142495902e-08f;
    p = p * z2 - 2.10102402082508e-06f;
    p = p * z2 - 5.69250639462346e-05f;
    p = p * z2 - 7.34990630326855e-04f;
    p = p * z2 - 2.95459980854025e-03f;
    p = p * z2 - 1.60960333262415e-02f;
    float q = -1.45660718464996e-05f;
    q = q * z2 - 2.13374055278905e-04f;
    q = q * z2 - 1.68282697438203e-03f;
    q = q * z2 - 7.37332916720468e-03f;
    q = q * z2 - 1.42647390514189e-02f;
    const float erf = (p * z) * __builtin_amdgcn_rcpf(q);
    return 0.5f * x * (1.0f + erf);
}

// XCD-aware block remap (bijective for any grid size): blocks that share an XCD
// (equal blockIdx % 8 under round-robin dispatch) get a contiguous range of
// logical tile ids, so neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace wm
