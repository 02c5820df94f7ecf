// Shared device helpers for the gfx950 kernels.  CDNA4 only: wave64, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
    static __device__ __forceinline__ elem from_f32_bounded(float x) { return (__bf16)x; }
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
        // saturate instead of producing inf: fp16 max is 65504.  One v_med3_f32 (fminf(fmaxf()) costs a NaN-canonicalising
        // v_max in front of it: 2.5 instead of 1.5 vector instructions per converted element); a NaN comes out as -65504 either way
        return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    }
    // for values known to lie inside fp16's range (softmax probabilities <= 2^6, convex combinations of 16-bit values):
    // no clamp, 0.5 instructions per element (v_cvt_pk_f16_f32)
    static __device__ __forceinline__ elem from_f32_bounded(float x) { return (_Float16)x; }
    static __device__ __forceinline__ float to_f32(elem x) { return (float)x; }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// e4m3 (OCP fp8, gfx950's native fp8) activations of WM_PREC_FP8: unit scale, round to nearest even, saturating at +-448
// (e4m3fn has no infinity).  4 floats -> 4 packed bytes.
struct FP8 {};
__device__ __forceinline__ unsigned pack4_e4m3(f32x4 v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], -448.0f, 448.0f);
    unsigned r = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0u, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], r, true);
}

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

// Column -> position inside a row of the fp8 blocks' stream planes (gemm8.h PLANES): each 256-column block holds its columns in the
// order the residual epilogue's passes touch them, [half ni][strip][32]: col = 64 strip + 32 ni + c  ->  pos = 128 ni + 32 strip + c.
// A bijection on aligned groups of 32 columns; col must be a multiple of 4 for the vector accesses that use it.
__host__ __device__ __forceinline__ int plane_pos(int col) {
    const int b = col & 255;
    return (col & ~255) + ((b >> 5) & 1) * 128 + (b >> 6) * 32 + (b & 31);
}
__host__ __device__ __forceinline__ int plane_col(int pos) {      // the inverse
    const int b = pos & 255;
    return (pos & ~255) + ((b >> 5) & 3) * 64 + (b >> 7) * 32 + (b & 31);
}

// exact-erf GELU of common.py:26 (nn.GELU default).  Two forms:
//   gelu_erf       - libm erff, used where the result stays fp32 (gemm32);
//   gelu_erf_fast  - erf(x / sqrt 2) as a clamped rational x P(x^2) / Q(x^2), degree 3/3 in x^2, |x| clamped to
//                    3.2 sqrt 2 (1 - erf(3.2) = 6e-6).  Least-squares fit weighted for the GELU error (coefficients
//                    from a host-side fit against scipy.special.erf, evaluated in fp32 in this operation order):
//                    |GELU error| <= 9.6e-6 for |x| <= 6 and <= 1.7e-6 |x| beyond; erf error <= 3.2e-6.
//                    Used in the 16-bit GEMM epilogues, whose output is rounded to 2^-9 / 2^-12 relative anyway.
//   gelu_erf_fast2 - the same on two values with packed fp32 math (v_pk_fma_f32) and ONE v_rcp_f32 for both
//                    (1 / (Qa Qb) times the other Q): the epilogue of the MLP's first GEMM is VALU-bound on this.
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
constexpr float GELU_X = 4.525483399593905f;
constexpr float GELU_P0 = 0.7978911995887756f, GELU_P1 = 0.05407121405005455f, GELU_P2 = 0.007685498800128698f,
                GELU_P3 = 6.773129280190915e-05f;
constexpr float GELU_Q1 = 0.2344742864370346f, GELU_Q2 = 0.02365594170987606f, GELU_Q3 = 0.0011780407512560487f;
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -GELU_X, GELU_X);
    const float t = xc * xc;
    float p = GELU_P3 * t + GELU_P2;
    p = p * t + GELU_P1;
    p = p * t + GELU_P0;
    float q = GELU_Q3 * t + GELU_Q2;
    q = q * t + GELU_Q1;
    q = q * t + 1.0f;
    const float e = (xc * p) * __builtin_amdgcn_rcpf(q);
    const float hx = 0.5f * x;
    return hx * e + hx;
}
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -GELU_X, GELU_X), __builtin_amdgcn_fmed3f(x[1], -GELU_X, GELU_X)};
    const f32x2 t = xc * xc;
    const f32x2 one = {1.0f, 1.0f};
    f32x2 p = __builtin_elementwise_fma(t, f32x2{GELU_P3, GELU_P3}, f32x2{GELU_P2, GELU_P2});
    p = __builtin_elementwise_fma(p, t, f32x2{GELU_P1, GELU_P1});
    p = __builtin_elementwise_fma(p, t, f32x2{GELU_P0, GELU_P0});
    f32x2 q = __builtin_elementwise_fma(t, f32x2{GELU_Q3, GELU_Q3}, f32x2{GELU_Q2, GELU_Q2});
    q = __builtin_elementwise_fma(q, t, f32x2{GELU_Q1, GELU_Q1});
    q = __builtin_elementwise_fma(q, t, one);
    const float r = __builtin_amdgcn_rcpf(q[0] * q[1]);           // Q in [1, 26]: the product cannot overflow
    const f32x2 rq = f32x2{q[1], q[0]} * f32x2{r, r};
    const f32x2 e = (xc * p) * rq;
    const f32x2 hx = x * f32x2{0.5f, 0.5f};
    return __builtin_elementwise_fma(hx, e, hx);
}
__device__ __forceinline__ f32x4 gelu_erf_fast4(f32x4 v) {
    const f32x2 a = gelu_erf_fast2(f32x2{v[0], v[1]}), b = gelu_erf_fast2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}
// GELU for an e4m3 OUTPUT (gemm8.h, lin1 of the fp8 blocks; round 4): x / (1 + 2^(-x (a + b x^2))), the tanh form written as a
// logistic, a and b refitted against the erf form (max |error| 2.7e-4 over the reals; e4m3 steps are 2^-4 relative: 1.8 % of the
// outputs land one step away and the rms distance to the exact value is unchanged, 2.64 %).  5 packed ops + 2 exp2 + 2 rcp per
// pair against 15 + 1 for gelu_erf_fast2.  No clamp: the exponent saturates to +-inf, the quotient to x or -0.
constexpr float GELU_SA = -2.3087653f, GELU_SB = -0.10012561f;
__device__ __forceinline__ f32x2 gelu_e4m3_fast2(f32x2 x) {
    const f32x2 t = x * x;
    const f32x2 w = __builtin_elementwise_fma(t, f32x2{GELU_SB, GELU_SB}, f32x2{GELU_SA, GELU_SA});
    const f32x2 z = x * w;
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + f32x2{1.0f, 1.0f};
    return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x4 gelu_e4m3_fast4(f32x4 v) {
    const f32x2 a = gelu_e4m3_fast2(f32x2{v[0], v[1]}), b = gelu_e4m3_fast2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}

// ---------------------------------------------------------------------------
// LayerNorm statistics in column tiles (shared by the fused GEMM epilogue, gemm16_v5.h, and layernorm_tiled_kernel,
// misc_kernels.h, so that both give bit-identical results whichever of them a batch size selects).  A row of N = TN * BN
// columns is TN partials; a partial is owned by a 16-lane group, lane l16 holding the 16-byte chunks l16 + 16 kk:
// two-pass (mean, M2) inside the partial, then Chan's combination of the TN equal-sized partials.
// ---------------------------------------------------------------------------
template <int CPT>
__device__ __forceinline__ void ln_partial16(const f32x4 (&v)[CPT], float inv_bn, float& mean, float& m2) {
    float s1 = 0.f;
#pragma unroll
    for (int kk = 0; kk < CPT; ++kk) s1 += (v[kk][0] + v[kk][1]) + (v[kk][2] + v[kk][3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    mean = s1 * inv_bn;
    m2 = 0.f;
#pragma unroll
    for (int kk = 0; kk < CPT; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[kk][j] - mean;
            m2 = fmaf(d, d, m2);
        }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
}
// mk / qk: the partials' means and M2s (ntile <= 8 used); returns the row mean, sets rstd
__device__ __forceinline__ float ln_combine(const float (&mk)[8], const float (&qk)[8], int ntile, float bn, float n, float eps, float& rstd) {
    float msum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < ntile) msum += mk[i];
    const float mean = msum / (float)ntile;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < ntile) {
            const float d = mk[i] - mean;
            m2 += fmaf(bn * d, d, qk[i]);
        }
    rstd = 1.0f / sqrtf(m2 / n + eps);
    return mean;
}
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float g, float b) {
    return fmaf((v - mean) * rstd, g, b);
}

// Element index of (row, col) of a [rows][C] 16-bit operand stored in LDS-image order (gemm16_v5.h "Operand layout"):
// [rows / 16][C / 32][64 positions x 8 elements], position = (row % 16) * 4 + (chunk ^ ((-((row % 16) >> 2)) & 3)),
// chunk = (col % 32) / 8.  rows % 16 == 0, C % 32 == 0.
__device__ __forceinline__ int64_t lds_image_index(int64_t row, int col, int C) {
    const int r = (int)(row & 15), cw = col & 31;
    const int pos = r * 4 + ((cw >> 3) ^ ((0 - (r >> 2)) & 3));
    return ((row >> 4) * (C >> 5) + (col >> 5)) * 512 + pos * 8 + (cw & 7);
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
