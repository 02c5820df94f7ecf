// 16-bit (bf16 / fp16) MFMA GEMM with fused epilogue, gfx950.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[(m % res_mod), N])
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights as
// stored), so A and W tiles are staged the same way.  This is the kernel behind
// every Linear / 1x1 conv / patch-embed GEMM on the path:
//   image_encoder.py:249,260 (qkv, proj), common.py:26 (lin1+GELU, lin2),
//   image_encoder.py:409-417,442-450 (patch / HFC embed as GEMM over patches),
//   image_encoder.py:494-513 (HFC adaptor projections), :105-121 (neck 1x1).
//
// Tiling: 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
// 4x4 MFMA 16x16x32 tiles), A/W tiles staged by LDS-DMA (global_load_lds, 16 B
// per lane) into an XOR-swizzled image -- the swizzle is applied to the per-lane
// SOURCE address and to the ds_read address, the LDS destination stays
// lane-linear (cdna_hip_programming.md §5.4 rule 21).  Two LDS buffers; the
// next K-tile's DMA is issued before the current tile's MFMAs.
// MFMA operand roles are swapped (W fragment as A-operand, activation fragment
// as B-operand) so each lane ends up with 4 consecutive N for one M row and the
// epilogue stores 16 B (fp32) / 8 B (16-bit) per lane.
#pragma once
#include "wm_common.h"

namespace wm {

constexpr int G16_BM = 128, G16_BN = 128, G16_BK = 64;
constexpr int G16_LDS_BYTES = 2 * (G16_BM + G16_BN) * G16_BK * 2;   // 64 KiB
constexpr int G16_GROUP_M = 8;

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };

struct Gemm16Args {
    const u16* A;
    const u16* W;
    const float* bias;       // [N] or null
    const float* residual;   // [res_mod, N] fp32 or null
    float* out32;            // [M,N] or null
    u16* out16;              // [M,N] or null
    int M, N, K;
    int res_mod;             // rows of residual (M, or 4096 for a per-tile broadcast)
    int act;
    // implicit-GEMM A operand (gemm16_v3.h, AMODE 1): A is an NHWC activation [M = B*64*64, conv_c] and the
    // GEMM's K runs over (tap, channel) of a 3x3 / pad 1 convolution, K = 9 * conv_c; out-of-image taps read
    // `zero_page` (>= 64 B of zeros).  Unused (0 / null) for a plain A matrix.
    int conv_c;
    const u16* zero_page;
    // row tiles per group of the grouped tile order (gemm16_v5.h; 0 = G16_GROUP_M): a group's row tiles x all column tiles
    // are consecutive tile ids, so group_m * tilesN ~ the 32 workgroups co-resident on an XCD keeps each A panel to one XCD
    int group_m;
    // gemm16_v5.h only: W / A stored in LDS-image order ([rows / 16][K / 32][64 x 16 B], pack16_lds_image_kernel); out_packed:
    // the 16-bit output is written in that order (it is the next GEMM's A operand; N % 32 == 0)
    int w_packed, a_packed, out_packed;
    // Folded LayerNorm (gemm16_v5.h "Folded LayerNorm").  Producer (FOLDP instance, fp32 + residual epilogue): st_stats
    // [M][N / BN][2] receives each row's (mean, M2) over this tile's columns, out16 the finished rows as 16-bit in LDS-image
    // order.  Consumer (16-bit epilogue): A is such a 16-bit copy x16 and W = gamma (.) W; with fold_stats = the producer's
    // partials over fold_ntile tiles of fold_bn columns (fold_ntile * fold_bn = K), fold_c1[n] = sum_k W[n][k] and
    // bias[n] = sum_k beta[k] W0[n][k] + b[n] the epilogue computes rstd (acc - mean c1) + bias = LayerNorm(x) W0^T + b.
    float* st_stats;
    const float* fold_stats;
    const float* fold_c1;
    int fold_ntile;
    float fold_bn, fold_eps;
    // Split residual stream (gemm16_v5.h "Split stream", round 4): the stream x as two 16-bit planes in LDS-image order,
    // hi = T(x) (the folded LayerNorm's operand) and lo = fp16(x - hi).  SPLIT instance: the residual comes in as
    // (res_hi, res_lo) and leaves as (out16 = hi, out_lo); the FOLDP instance (fp32 residual in) writes out_lo too when it is
    // given and then skips out32 when that is null.  overflow: a host-visible word the producers set to 1 when a value of the
    // stream reaches the fp16 clamp (|x| >= 65504), or null.
    const u16* res_hi;
    const u16* res_lo;
    u16* out_lo;
    int* overflow;
};

template <class T>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int tilesM = p.M / G16_BM, tilesN = p.N / G16_BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    // grouped order: G16_GROUP_M row tiles share each W panel back to back
    const int per_group = G16_GROUP_M * tilesN;
    const int group = lid / per_group;
    const int first_m = group * G16_GROUP_M;
    const int gsz = min(G16_GROUP_M, tilesM - first_m);
    const int in_group = lid - group * per_group;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int m0 = tm * G16_BM, n0 = tn * G16_BN;
    const int K = p.K;
    const int nk = K / G16_BK;

    const char* Ab = (const char*)p.A;
    const char* Wb = (const char*)p.W;

    // per-lane source offsets for the 4 (A) + 4 (W) DMA pieces this wave issues per K-tile
    size_t a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int seg = wave * 4 + i;
        const int r = seg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        a_off[i] = ((size_t)(m0 + r) * K) * 2 + c * 16;
        w_off[i] = ((size_t)(n0 + r) * K) * 2 + c * 16;
    }

    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * 32768;
        char* sW = sA + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int seg = wave * 4 + i;
            __builtin_amdgcn_global_load_lds(Ab + a_off[i] + (size_t)kt * 128, WM_LDS_PTR(sA + seg * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(Wb + w_off[i] + (size_t)kt * 128, WM_LDS_PTR(sW + seg * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside a tile image, per k-step
    const int fr = lane & 15, fq = lane >> 4;
    int a_rd[4][2], w_rd[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + fr;
        const int rw = wc * 64 + i * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int q = ks * 4 + fq;
            a_rd[i][ks] = ra * 128 + ((q ^ (ra & 7)) << 4);
            w_rd[i][ks] = rw * 128 + ((q ^ (rw & 7)) << 4);
        }
    }

    stage(0, 0);
    __syncthreads();   // hipcc drains the DMA (vmcnt(0)) in front of the barrier

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sA = smem + cur * 32768;
        const char* sW = sA + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            typename T::vec8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *(const typename T::vec8*)(sA + a_rd[i][ks]);
                wf[i] = *(const typename T::vec8*)(sW + w_rd[i][ks]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = T::mfma16(wf[ni], af[mi], acc[mi][ni]);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
    const int res_mod = p.res_mod > 0 ? p.res_mod : p.M;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc * 64 + ni * 16 + fq * 4;
            f32x4 v = acc[mi][ni];
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + n);
                v += b;
            }
            if (p.act == ACT_GELU) {
                v = gelu_erf_fast4(v);      // the same arithmetic in every GEMM kernel: a tile's bits must not depend on which one its batch size selects
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (p.residual) {
                const f32x4 r = *(const f32x4*)(p.residual + (size_t)(m % res_mod) * p.N + n);
                v += r;
            }
            if (p.out32) *(f32x4*)(p.out32 + (size_t)m * p.N + n) = v;
            if (p.out16) {
                typename T::vec4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                *(typename T::vec4*)(p.out16 + (size_t)m * p.N + n) = o;
            }
        }
    }
}

}  // namespace wm
