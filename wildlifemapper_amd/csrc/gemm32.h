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

// ---------------------------------------------------------------------------
// The same GEMM on the 16-bit matrix pipe, fp32 in and out (round 4): every fp32 operand value x is split into two fp16 numbers,
// hi = fp16(x), lo = fp16(x - hi)  (x = hi + lo to 2^-22 |x|), and  A W^T ~= Ah Wh^T + Al Wh^T + Ah Wl^T  (the dropped Al Wl^T is
// 2^-22 of a product; fp16 x fp16 products are exact in the fp32 accumulator).  Three 32x32x16 MFMAs (96 cycles per 16 of K) stand for
// the eight 32x32x2 fp32 MFMAs (512 cycles) of gemm32_kernel; measured error against float64: rel-L2 3e-7 (gemm32_kernel: 1e-7; the
// test bound for both is 2e-6).  W is scaled by 2^6 before its split (exact; taken back in the epilogue) so that decoder-sized weights
// (|w| ~ 0.02) keep their lo part out of fp16's subnormals.  W comes pre-split (two fp16 planes made once per weight upload:
// `split_w32_kernel`) or as fp32 (split per K-step like A: the op-level entry).  |A| >= 65504 or |W| >= 1023 leaves fp16's range: the
// conversions do not clamp, so such a row comes out inf / nan, and the kernels raise the handle's overflow word (wm_stream_overflow)
// as the fp16 stream's producers do.
// Same tile (64 x 64 x 32, 4 waves as 2 x 2), same ragged M / N handling; K % 32 == 0.  Operand tiles are fetched two K-steps ahead
// (the decoder's token-side calls are 13 x 4 workgroups: the load latency is all there is).
// ---------------------------------------------------------------------------
constexpr float G32X3_WSCALE = 64.0f;
struct Gemm32x3Args {
    const float* A; const float* W; const u16* Whi; const u16* Wlo;      // W (fp32) or (Whi, Wlo)
    const float* bias; const float* residual; float* out;
    int M, N, K, act, lda;
    int* overflow;
};

__global__ __launch_bounds__(256) void split_w32_kernel(const float* __restrict__ w, u16* __restrict__ hi, u16* __restrict__ lo, int64_t n4, int* overflow) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(w + i * 4) * G32X3_WSCALE;
        typename FP16::vec4 h, l;
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = FP16::from_f32_bounded(v[j]);                 // no clamp: out of range = inf, i.e. a loud result
            l[j] = FP16::from_f32_bounded(v[j] - FP16::to_f32(h[j]));
            amax = fmaxf(amax, fabsf(v[j]));
        }
        *(typename FP16::vec4*)(hi + i * 4) = h;
        *(typename FP16::vec4*)(lo + i * 4) = l;
        if (amax >= 65504.f && overflow) *(volatile int*)overflow = 1;
    }
}

template <bool WPRE>
__global__ __launch_bounds__(256) void gemm32x3_kernel(Gemm32x3Args p) {
    constexpr int BM = 64, BN = 64, BK = 32, LDH = BK + 8;    // LDS row = 80 bytes: conflict-free 16-byte fragment reads
    __shared__ __attribute__((aligned(16))) u16 sAh[BM * LDH], sAl[BM * LDH], sWh[BN * LDH], sWl[BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;
    const int lrow = tid >> 2, lcol = (tid & 3) * 8;          // 8 consecutive k per thread and operand

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool arow = m0 + lrow < p.M, wrow = n0 + lrow < p.N;
    const float* ap = p.A + (size_t)(arow ? m0 + lrow : 0) * p.lda + lcol;
    const size_t woff = (size_t)(wrow ? n0 + lrow : 0) * p.K + lcol;
    struct Stage { f32x4 a0, a1, w0, w1; uint4 wh, wl; };
    Stage st[2];
    float amax = 0.f;
    auto fetch = [&](Stage& s, int k0) {
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        s.a0 = s.a1 = s.w0 = s.w1 = zero;
        s.wh = s.wl = uint4{0u, 0u, 0u, 0u};
        if (arow) { s.a0 = *(const f32x4*)(ap + k0); s.a1 = *(const f32x4*)(ap + k0 + 4); }
        if (wrow) {
            if constexpr (WPRE) { s.wh = *(const uint4*)(p.Whi + woff + k0); s.wl = *(const uint4*)(p.Wlo + woff + k0); }
            else { s.w0 = *(const f32x4*)(p.W + woff + k0); s.w1 = *(const f32x4*)(p.W + woff + k0 + 4); }
        }
    };
    auto split8 = [&](const f32x4& x0, const f32x4& x1, float scale, uint4& hi, uint4& lo) {
        union { uint4 raw; typename FP16::vec4 h[2]; } uh, ul;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (j < 4 ? x0[j] : x1[j - 4]) * scale;
            amax = fmaxf(amax, fabsf(x));
            uh.h[j >> 2][j & 3] = FP16::from_f32_bounded(x);     // no clamp: |x| >= 65520 becomes inf and the row comes out nan / inf
            ul.h[j >> 2][j & 3] = FP16::from_f32_bounded(x - FP16::to_f32(uh.h[j >> 2][j & 3]));
        }
        hi = uh.raw; lo = ul.raw;
    };
    const int nk = p.K / BK;
    fetch(st[0], 0);
    if (nk > 1) fetch(st[1], BK);
    const int fa = (wr * 32 + (lane & 31)) * LDH + 8 * (lane >> 5), fw = (wc * 32 + (lane & 31)) * LDH + 8 * (lane >> 5);
    auto step = [&](Stage& s, int kt) {
        uint4 ah, al, wh, wl;
        split8(s.a0, s.a1, 1.0f, ah, al);
        if constexpr (WPRE) { wh = s.wh; wl = s.wl; }
        else split8(s.w0, s.w1, G32X3_WSCALE, wh, wl);
        __syncthreads();                                      // the previous step's fragment reads are done
        *(uint4*)(sAh + lrow * LDH + lcol) = ah; *(uint4*)(sAl + lrow * LDH + lcol) = al;
        *(uint4*)(sWh + lrow * LDH + lcol) = wh; *(uint4*)(sWl + lrow * LDH + lcol) = wl;
        __syncthreads();
        if (kt + 2 < nk) fetch(s, (kt + 2) * BK);             // this stage's registers are free again: two K-steps ahead
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const typename FP16::vec8 a_hi = *(const typename FP16::vec8*)(sAh + fa + 16 * ks), a_lo = *(const typename FP16::vec8*)(sAl + fa + 16 * ks);
            const typename FP16::vec8 w_hi = *(const typename FP16::vec8*)(sWh + fw + 16 * ks), w_lo = *(const typename FP16::vec8*)(sWl + fw + 16 * ks);
            acc = FP16::mfma32(a_lo, w_hi, acc);              // the two small terms first
            acc = FP16::mfma32(a_hi, w_lo, acc);
            acc = FP16::mfma32(a_hi, w_hi, acc);
        }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(st[0], kt);
        if (kt + 1 < nk) step(st[1], kt + 1);
    }
    if (amax >= 65504.f && p.overflow) *(volatile int*)p.overflow = 1;

    const int n = n0 + wc * 32 + (lane & 31);
    if (n >= p.N) return;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < p.M) {
            float v = acc[r] * (1.0f / G32X3_WSCALE) + bv;
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_GELU) v = gelu_erf(v);
            else if (p.act == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            if (p.residual) v += p.residual[(size_t)m * p.N + n];
            p.out[(size_t)m * p.N + n] = v;
        }
    }
}

}  // namespace wm
