// fp32 GEMM on the fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact f32 fmaf chains.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])
//
// Used for the whole detection decoder (box_decoder.py:71-149, transformer.py),
// which is 3.55 GFLOP/tile (0.06 % of the path) but amplifies any error in the
// embedding into the logits, so it is kept in fp32 end to end.  Ragged M and N
// are allowed (51 tokens per tile, 8 / 4 head outputs); K must be a multiple of 16.
// 64x64x16 tile per 256-thread workgroup, 4 waves as 2x2, one 32x32 MFMA tile each.
#pragma once
#include "wm_common.h"
#include "gemm16.h"   // ACT_* enum

namespace wm {

struct Gemm32Args {
    const float* A; const float* W; const float* bias; const float* residual; float* out;
    int M, N, K, act;
    int lda;   // row stride of A in floats (>= K)
};

__global__ __launch_bounds__(256) void gemm32_kernel(Gemm32Args p) {
    constexpr int BM = 64, BN = 64, BK = 16, LD = BK + 1;
    __shared__ float sA[BM * LD];
    __shared__ float sW[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lrow = tid >> 2, lcol = (tid & 3) * 4;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int k0 = 0; k0 < p.K; k0 += BK) {
        f32x4 va = f32x4{0.f, 0.f, 0.f, 0.f}, vw = va;
        if (m0 + lrow < p.M) va = *(const f32x4*)(p.A + (size_t)(m0 + lrow) * p.lda + k0 + lcol);
        if (n0 + lrow < p.N) vw = *(const f32x4*)(p.W + (size_t)(n0 + lrow) * p.K + k0 + lcol);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sA[lrow * LD + lcol + j] = va[j];
            sW[lrow * LD + lcol + j] = vw[j];
        }
        __syncthreads();
        const float* pa = sA + (wr * 32 + (lane & 31)) * LD + (lane >> 5);
        const float* pw = sW + (wc * 32 + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], pw[kk], acc, 0, 0, 0);
    }

    const int n = n0 + wc * 32 + (lane & 31);
    if (n >= p.N) return;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < p.M) {
            float v = acc[r] + bv;
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_GELU) v = gelu_erf(v);
            else if (p.act == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            if (p.residual) v += p.residual[(size_t)m * p.N + n];
            p.out[(size_t)m * p.N + n] = v;
        }
    }
}

}  // namespace wm
