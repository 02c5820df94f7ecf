// fp32 GEMM on the fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact f32 fmaf chains.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])
//
// Used for the whole detection decoder (box_decoder.py:71-149, transformer.py),
// which is 3.55 GFLOP/tile (0.06 % of the path) but amplifies any error in the
// embedding into the logits, so it is kept in fp32 end to end.  Ragged M and N
// are allowed (51 tokens per tile, 8 / 4 head outputs); K must be a multiple of 16.
// 64x64x32 tile per 256-thread workgroup, 4 waves as 2x2, one 32x32 MFMA tile each.
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
    constexpr int BM = 64, BN = 64, BK = 32, LD = BK + 1;
    __shared__ float sA[BM * LD];
    __shared__ float sW[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // 1-D grid, XCD-aware: the column tiles of one row block read the same A rows, so they get consecutive logical ids = one XCD's
    // L2 (as a 2-D grid they were dealt round-robin over the XCDs and A came from HBM once per column tile: the decoder's
    // M = 65536-row projections, N = 128 / 256)
    const int ntn = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;
    const int lrow = tid >> 2, lcol = (tid & 3) * 8;          // 8 consecutive k per thread and operand

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // the next K-tile is fetched into registers while the MFMAs of the current one run (the decoder's token-side
    // GEMMs have M = 51 tokens per tile: a handful of workgroups, so the load latency is what there is to hide)
    const bool arow = m0 + lrow < p.M, wrow = n0 + lrow < p.N;
    const float* ap = p.A + (size_t)(arow ? m0 + lrow : 0) * p.lda + lcol;
    const float* wp = p.W + (size_t)(wrow ? n0 + lrow : 0) * p.K + lcol;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 va0 = zero, va1 = zero, vw0 = zero, vw1 = zero;
    auto fetch = [&](int k0) {
        const bool kin = k0 + lcol < p.K;                     // K % 16 == 0: the last tile may be half empty
        va0 = va1 = vw0 = vw1 = zero;
        if (arow && kin) { va0 = *(const f32x4*)(ap + k0); va1 = *(const f32x4*)(ap + k0 + 4); }
        if (wrow && kin) { vw0 = *(const f32x4*)(wp + k0); vw1 = *(const f32x4*)(wp + k0 + 4); }
    };
    fetch(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sA[lrow * LD + lcol + j] = va0[j]; sA[lrow * LD + lcol + 4 + j] = va1[j];
            sW[lrow * LD + lcol + j] = vw0[j]; sW[lrow * LD + lcol + 4 + j] = vw1[j];
        }
        __syncthreads();
        if (k0 + BK < p.K) fetch(k0 + BK);
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
