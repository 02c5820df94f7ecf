// HBM-bound helper kernels: LayerNorm, patch gather, transposes, converts.
#pragma once
#include <type_traits>

#include "wm_common.h"

namespace wm {

// ---------------------------------------------------------------------------
// LayerNorm over the last dim of [rows, C] fp32, biased variance, two-pass in
// registers (nn.LayerNorm / LayerNorm2d per pixel: image_encoder.py:173,183,
// common.py:31-43).  One wave per row, C = 256*NV, 16 B per lane per step.
// Outputs: fp32 [rows,C] and/or 16-bit [rows,C]; or, if nchw_hw > 0, fp32 in
// channel-major order out[(row / hw), c, row % hw] (neck output, NHWC -> NCHW).
// ---------------------------------------------------------------------------
template <class T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* out32, u16* out16, int64_t rows, int nchw_hw) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(sum) * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            sq += d * d;
        }
    const float var = wave_sum(sq) * (1.0f / C);
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = i * 256 + lane * 4;
        const f32x4 g = *(const f32x4*)(gamma + c0);
        const f32x4 b = *(const f32x4*)(beta + c0);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
        if (nchw_hw > 0) {
            const int64_t img = row / nchw_hw, pix = row % nchw_hw;
#pragma unroll
            for (int j = 0; j < 4; ++j) out32[(img * C + c0 + j) * nchw_hw + pix] = y[j];
        } else {
            if (out32) *(f32x4*)(out32 + row * C + c0) = y;
            if (out16) {
                typename T::vec4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = T::from_f32(y[j]);
                *(typename T::vec4*)(out16 + row * C + c0) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The transformer blocks' LayerNorm with the column-tiled statistics of wm_common.h (ln_partial16 / ln_combine): the
// same arithmetic, operation for operation, as the LayerNorm fused into the residual GEMMs (gemm16_v5.h), which only
// some batch sizes can use -- a tile's result must not depend on its batch neighbours.  One wave per row, 16-lane
// group k owns column tile k (C = TN * BN, TN <= 4), 16-bit output.  grid (rows / 4), 256 threads.
// ---------------------------------------------------------------------------
// store 4 consecutive outputs of type T (16-bit) or, for FP8, as packed e4m3 bytes (wm_common.h)
template <class T>
__device__ __forceinline__ void store4_as(void* base, int64_t elem, const f32x4& y) {
    if constexpr (std::is_same<T, FP8>::value) {
        *(unsigned*)((unsigned char*)base + elem) = pack4_e4m3(y);
    } else {
        typename T::vec4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = T::from_f32(y[j]);
        *(typename T::vec4*)((u16*)base + elem) = o;
    }
}

// row-major [rows][K] 16-bit -> LDS-image order (weights at wm_finalize_weights; tests).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void pack16_lds_image_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t rows, int K) {
    const int64_t n16 = rows * (K / 8);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / (K / 8);
        const int col = (int)(i - row * (K / 8)) * 8;
        out[lds_image_index(row, col, K) >> 3] = in[i];
    }
}

template <class T, int BN>
__global__ __launch_bounds__(256) void layernorm_tiled_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, u16* __restrict__ out16,
                                                              int64_t rows, int C, int packed) {
    constexpr int CPT = BN / 64;                       // 16-byte chunks per lane (BN / 4 per tile over 16 lanes)
    const int lane = threadIdx.x & 63, k = lane >> 4, l16 = lane & 15;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int ntile = C / BN;
    const bool live = k < ntile;
    const float* xr = x + row * C + k * BN;
    f32x4 v[CPT];
#pragma unroll
    for (int kk = 0; kk < CPT; ++kk) v[kk] = live ? *(const f32x4*)(xr + (l16 + 16 * kk) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    float pm, pq;
    ln_partial16<CPT>(v, 1.0f / BN, pm, pq);
    float mk[8], qk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mk[i] = i < 4 ? __shfl(pm, i * 16, 64) : 0.f;
        qk[i] = i < 4 ? __shfl(pq, i * 16, 64) : 0.f;
    }
    float rstd;
    const float mean = ln_combine(mk, qk, ntile, (float)BN, (float)C, eps, rstd);
    if constexpr (!std::is_same<T, FP8>::value) {
        if (packed) {
            // The output is the A operand of a gemm16_v5 launch: LDS-image order (gemm16_v5.h "Operand layout").  A row's 64 bytes
            // of one K-step are contiguous there and the workgroup's 4 consecutive rows make 256 B, so the rows go through LDS and
            // the workgroup stores 16 bytes per thread, 256-byte runs (a wave storing its own row would write 64-byte pieces 1 KiB
            // apart: measured +15 % on this HBM-bound kernel).  rows % 16 == 0: every workgroup is full.
            __shared__ __attribute__((aligned(16))) u16 stage[4][1280];
            const int w = threadIdx.x >> 6;
            if (live) {
#pragma unroll
                for (int kk = 0; kk < CPT; ++kk) {
                    const int c0 = k * BN + (l16 + 16 * kk) * 4;
                    const f32x4 g = *(const f32x4*)(gamma + c0);
                    const f32x4 b = *(const f32x4*)(beta + c0);
                    typename T::vec4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = T::from_f32(ln_apply(v[kk][j], mean, rstd, g[j], b[j]));
                    *(typename T::vec4*)(&stage[w][c0]) = o;
                }
            }
            __syncthreads();
            const int64_t row0 = (int64_t)blockIdx.x * 4;
            const int r0 = (int)(row0 & 15);                               // 0, 4, 8 or 12: the 4 rows share r >> 2, i.e. one swizzle key
            const int key = (0 - (r0 >> 2)) & 3;
            uint4* dst = (uint4*)(out16 + (row0 >> 4) * (int64_t)(C >> 5) * 512) + r0 * 4;      // + kt * 64 + (row in 4) * 4 + position chunk
            for (int i = threadIdx.x; i < (C >> 5) * 16; i += 256) {       // per K-step: 4 rows x 4 chunks = 256 contiguous bytes
                const int kt = i >> 4, rr = (i >> 2) & 3, pc = i & 3;
                dst[kt * 64 + rr * 4 + pc] = *(const uint4*)(&stage[rr][kt * 32 + ((pc ^ key) << 3)]);
            }
            return;
        }
    }
    if (!live) return;
#pragma unroll
    for (int kk = 0; kk < CPT; ++kk) {
        const int c0 = k * BN + (l16 + 16 * kk) * 4;
        const f32x4 g = *(const f32x4*)(gamma + c0);
        const f32x4 b = *(const f32x4*)(beta + c0);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = ln_apply(v[kk][j], mean, rstd, g[j], b[j]);
        store4_as<T>(out16, row * C + c0, y);
    }
}

// The fp8 blocks' LayerNorm on the stream's hi plane (gemm8.h PLANES): position-wise.  Input: the hi plane of rows (16-bit type TIN,
// column c at plane_pos(c)); output: e4m3 rows IN THE SAME ORDER (position p holds LN(x)[plane_col(p)]), consumed by a gemm8 launch whose
// weight has its K columns permuted alike (wm_api.hip upload8: a dot product does not care).  So the kernel reads and writes whole
// contiguous runs: a lane owns the 16-byte chunks lane + 64 j of the row (1 KiB per load instruction, 512 B per store instruction;
// the column-tiled kernel above wrote 64-byte pieces 320 columns apart).  Statistics: mean, then the centred second moment, both over
// the wave.  One wave per row, NJ = ceil(C / 512) chunks per lane, grid (rows / 4) x 256 threads.
template <class TIN, int NJ, int RPW>
__global__ __launch_bounds__(256) void layernorm_plane_fp8_kernel(const u16* __restrict__ hi, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float eps, unsigned char* __restrict__ out, int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;       // RPW rows per wave, all loads issued before the first use (1 and 2 measured equal: 1)
    if (row0 >= rows) return;
    const int nch = C >> 3;
    uint4 raw[RPW][NJ];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lane + 64 * j < nch && row0 + r < rows) raw[r][j] = *(const uint4*)(hi + (row0 + r) * C + (lane + 64 * j) * 8);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
        float v[NJ][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lane + 64 * j < nch) {
                union { uint4 raw; typename TIN::vec4 h[2]; } u;
                u.raw = raw[r][j];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[j][e] = TIN::to_f32(u.h[e >> 2][e & 3]);
                    s += v[j][e];
                }
            }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lane + 64 * j < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[j][e] -= mean;
                    q = fmaf(v[j][e], v[j][e], q);
                }
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = lane + 64 * j;
            if (p < nch) {
                const int c0 = plane_col(p * 8);              // 8 consecutive positions are 8 consecutive columns
                uint2 o;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 g = *(const f32x4*)(gamma + c0 + 4 * hh);
                    const f32x4 b = *(const f32x4*)(beta + c0 + 4 * hh);
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = fmaf(v[j][4 * hh + e] * rstd, g[e], b[e]);
                    (hh ? o.y : o.x) = pack4_e4m3(y);
                }
                *(uint2*)(out + row * C + p * 8) = o;
            }
        }
    }
}

// the stream's planes of rows -> fp16 rows in column order (the neck's operand): out[row][c] = fp16(float(hi) + float(lo))
template <class T>
__global__ __launch_bounds__(256) void stream_rows_to_fp16_kernel(const u16* __restrict__ hi, const u16* __restrict__ lo, u16* __restrict__ out, int64_t n4, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i * 4 / C;
        const int64_t at = row * C + plane_pos((int)(i * 4 - row * C));
        const typename T::vec4 h4 = *(const typename T::vec4*)(hi + at);
        const f16x4 l4 = *(const f16x4*)(lo + at);
        typename FP16::vec4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = FP16::from_f32(T::to_f32(h4[j]) + (float)l4[j]);
        *(typename FP16::vec4*)(out + i * 4) = o;
    }
}

// Folded LayerNorm, weight side (once per wm_finalize_weights): from the weight W [N][K] (row-major), the LayerNorm's
// gamma / beta [K] and the Linear's bias [N]:
//   wf[n][k] = round16(gamma[k] * W[n][k])   written in LDS-image order
//   c1[n] = sum_k wf[n][k]        c2[n] = sum_k beta[k] * W[n][k] + bias[n]
// W is the fp32 weight when `w32` is given (the engine's path since round 4: the folded weight is then rounded ONCE, which is what
// lets bf16-operand blocks fold: rounded twice, bf16 cost 8.7-9.5e-4 against 7.2-8.2e-4 on the ViT-H logits), else the 16-bit
// weight `w16` as packed for the unfolded path (wm_op_fold_weight16: the op-level identity tests against LayerNorm-then-GEMM).
// One workgroup per output row; fixed summation order (thread-strided partials, then a fixed tree): the same bits every time.
template <class T>
__global__ __launch_bounds__(256) void fold_weight_kernel(const u16* __restrict__ w16, const float* __restrict__ w32, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ bias, u16* __restrict__ wf,
                                                          float* __restrict__ c1, float* __restrict__ c2, int N, int K) {
    const int n = blockIdx.x, tid = threadIdx.x;
    __shared__ float red[2][256];
    float s1 = 0.f, s2 = 0.f;
    for (int k4 = tid; k4 < K / 4; k4 += 256) {
        f32x4 wv;
        if (w32) {
            wv = *(const f32x4*)(w32 + (size_t)n * K + k4 * 4);
        } else {
            const typename T::vec4 w = *(const typename T::vec4*)(w16 + (size_t)n * K + k4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = T::to_f32(w[j]);
        }
        const f32x4 g = *(const f32x4*)(gamma + k4 * 4), b = *(const f32x4*)(beta + k4 * 4);
        typename T::vec4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = T::from_f32(g[j] * wv[j]);
            s1 += T::to_f32(o[j]);
            s2 = fmaf(b[j], wv[j], s2);
        }
        *(typename T::vec4*)(wf + lds_image_index(n, k4 * 4, K)) = o;
    }
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { c1[n] = red[0][0]; c2[n] = red[1][0] + (bias ? bias[n] : 0.f); }
}

// Folded LayerNorm (gemm16_v5.h "Folded LayerNorm"), standalone producer: per-row partial statistics over BN-column tiles,
// [rows][C / BN][2] = (mean, M2) -- the same arithmetic, lane assignment and order as the residual GEMM's FOLDP epilogue and
// as layernorm_tiled_kernel above -- and the 16-bit copy of the rows in LDS-image order.  Used where the residual stream was
// not produced by a FOLDP launch: a half-width GEMM (one or two tiles per call), or an operand-type boundary between blocks.
// Split stream (round 4, gemm16_v5.h "Split stream"): with `lo16` the kernel also writes lo = fp16(x - hi) in the same order, and with
// `x_rw` it rewrites the fp32 rows as float(hi) + float(lo) -- the value the SPLIT GEMM instance reconstructs -- so that the
// half-width path (fp32 stream) and the split path carry the same stream bit for bit.  overflow: set to 1 when an fp16 hi clamps.
template <class T, int BN>
__global__ __launch_bounds__(256) void ln_stats_x16_kernel(const float* __restrict__ x, float* __restrict__ stats, u16* __restrict__ x16,
                                                           int64_t rows, int C, u16* __restrict__ lo16 = nullptr, float* x_rw = nullptr,
                                                           int* overflow = nullptr) {
    constexpr int CPT = BN / 64;
    const int lane = threadIdx.x & 63, k = lane >> 4, l16 = lane & 15;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int ntile = C / BN;
    const bool live = k < ntile;
    const float* xr = x + row * C + k * BN;
    f32x4 v[CPT];
#pragma unroll
    for (int kk = 0; kk < CPT; ++kk) v[kk] = live ? *(const f32x4*)(xr + (l16 + 16 * kk) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    float pm, pq;
    ln_partial16<CPT>(v, 1.0f / BN, pm, pq);
    if (!live) return;
    if (l16 == 0) *(float2*)(stats + (row * ntile + k) * 2) = make_float2(pm, pq);
    float amax = 0.f;
#pragma unroll
    for (int kk = 0; kk < CPT; ++kk) {
        typename T::vec4 o;
        f16x4 lo;
        f32x4 back;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = T::from_f32(v[kk][j]);
            lo[j] = FP16::from_f32(v[kk][j] - T::to_f32(o[j]));
            back[j] = T::to_f32(o[j]) + (float)lo[j];
            amax = fmaxf(amax, fabsf(v[kk][j]));
        }
        const int64_t e = lds_image_index(row, k * BN + (l16 + 16 * kk) * 4, C);
        *(typename T::vec4*)(x16 + e) = o;
        if (lo16) *(f16x4*)(lo16 + e) = lo;
        if (x_rw) *(f32x4*)(x_rw + row * C + k * BN + (l16 + 16 * kk) * 4) = back;
    }
    if constexpr (std::is_same<T, FP16>::value) {
        if (amax >= 65504.f && overflow) *(volatile int*)overflow = 1;
    }
}

// Split stream -> fp32 rows: out[row][col] = float(hi) + float(lo), planes in LDS-image order (taps, the neck's input in bf16 mode,
// operand-type boundaries between blocks).  One thread per 16-byte chunk of a plane (8 elements).
template <class T>
__global__ __launch_bounds__(256) void stream_merge_kernel(const u16* __restrict__ hi, const u16* __restrict__ lo, float* __restrict__ out,
                                                           int64_t rows, int C) {
    const int64_t n16 = rows * (C / 8);
    const int kts = C >> 5;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        // chunk i of the image: [row block][K-step][position], position = (row % 16) * 4 + swizzled chunk
        const int pos = (int)(i & 63);
        const int64_t piece = i >> 6;
        const int kt = (int)(piece % kts);
        const int64_t rb = piece / kts;
        const int r = pos >> 2, chunk = (pos & 3) ^ ((0 - (r >> 2)) & 3);
        const typename T::vec8 h = *(const typename T::vec8*)(hi + i * 8);
        const f16x8 l = *(const f16x8*)(lo + i * 8);
        float* dst = out + (rb * 16 + r) * C + kt * 32 + chunk * 8;
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = T::to_f32(h[j]) + (float)l[j];
            b[j] = T::to_f32(h[4 + j]) + (float)l[4 + j];
        }
        *(f32x4*)dst = a;
        *(f32x4*)(dst + 4) = b;
    }
}

// Row forms of the split stream (the fp8 blocks: gemm8.h PLANES; column c of a row sits at plane_pos(c)): fp32 rows -> (hi, lo);
// (hi, lo) -> fp32 rows.  n4 = elements / 4, C % 256 == 0.
template <class T>
__global__ __launch_bounds__(256) void stream_split_rows_kernel(const float* __restrict__ x, u16* __restrict__ hi, u16* __restrict__ lo, int64_t n4, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i * 4 / C;
        const int64_t at = row * C + plane_pos((int)(i * 4 - row * C));
        const f32x4 v = *(const f32x4*)(x + i * 4);
        typename T::vec4 o;
        f16x4 l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = T::from_f32(v[j]);
            l[j] = FP16::from_f32(v[j] - T::to_f32(o[j]));
        }
        *(typename T::vec4*)(hi + at) = o;
        *(f16x4*)(lo + at) = l;
    }
}
template <class T>
__global__ __launch_bounds__(256) void stream_merge_rows_kernel(const u16* __restrict__ hi, const u16* __restrict__ lo, float* __restrict__ out, int64_t n4, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i * 4 / C;
        const int64_t at = row * C + plane_pos((int)(i * 4 - row * C));
        const typename T::vec4 h4 = *(const typename T::vec4*)(hi + at);
        const f16x4 l4 = *(const f16x4*)(lo + at);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = T::to_f32(h4[j]) + (float)l4[j];
        *(f32x4*)(out + i * 4) = v;
    }
}

// LDS-image order -> row-major (the neck's first GEMM where it is a half-width launch: 1-4 tiles per call)
__global__ __launch_bounds__(256) void unpack16_lds_image_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t rows, int K) {
    const int64_t n16 = rows * (K / 8);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / (K / 8);
        const int col = (int)(i - row * (K / 8)) * 8;
        out[i] = in[lds_image_index(row, col, K) >> 3];
    }
}

// ---------------------------------------------------------------------------
// Batched 2-byte transpose: in [batch][R][C] -> out [batch][C][R], 64x64 LDS tiles.
// Used for the HFC adaptor's scramble reshape (image_encoder.py:512): per tile the
// [4096 tok, 1024 ch] buffer re-read as [1024, 4096] must become the K-contiguous
// A operand [4096, 1024] of proj_back.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose16_kernel(const u16* __restrict__ in, u16* __restrict__ out, int R, int C) {
    __shared__ u16 tile[64][66];
    const int64_t boff = (int64_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty + i * 4;
        tile[r][tx] = in[boff + (int64_t)(r0 + r) * C + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = ty + i * 4;
        out[boff + (int64_t)(c0 + cc) * R + r0 + tx] = tile[tx][cc];
    }
}

// fp32 [batch][R][C] -> [batch][C][R] (NCHW <-> NHWC of the 256-channel embedding)
__global__ __launch_bounds__(256) void transpose32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[64][65];
    const int64_t boff = (int64_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty + i * 4;
        tile[r][tx] = in[boff + (int64_t)(r0 + r) * C + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = ty + i * 4;
        out[boff + (int64_t)(c0 + cc) * R + r0 + tx] = tile[tx][cc];
    }
}

template <class T>
__global__ __launch_bounds__(256) void cvt_f32_to_16_kernel(const float* __restrict__ in, u16* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(in + i * 4);
        typename T::vec4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
        *(typename T::vec4*)(out + i * 4) = o;
    }
}

// Opt-in saturation census (wm_debug_saturation_*): elements of a 16-bit buffer whose magnitude bits are >= `thr` (fp16:
// 0x7bff = 65504, the value T::from_f32 clamps to, and inf / NaN above it; bf16: 0x7f7f and above), or of an e4m3 byte
// buffer with magnitude >= 0x7e (448).  One atomic per workgroup.  16 bytes per thread and iteration.
template <int BYTES_PER_ELEM>
__global__ __launch_bounds__(256) void saturation_count_kernel(const uint4* __restrict__ in, int64_t n16, unsigned thr, unsigned long long* __restrict__ counter) {
    unsigned c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const uint4 v = in[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (BYTES_PER_ELEM == 2) {
                c += ((w[j] & 0x7fffu) >= thr) + (((w[j] >> 16) & 0x7fffu) >= thr);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) c += (((w[j] >> (8 * b)) & 0x7fu) >= thr);
            }
        }
    }
    c = (unsigned)wave_sum((float)c);                  // <= 64 * 16 * iterations: exact in fp32 for any realistic grid
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(counter, (unsigned long long)c);
}

template <class T>
__global__ __launch_bounds__(256) void cvt_16_to_f32_kernel(const u16* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const typename T::elem e = __builtin_bit_cast(typename T::elem, in[i]);
        out[i] = T::to_f32(e);
    }
}

// ---------------------------------------------------------------------------
// Input pipeline: uint8 HWC image (h, w <= 1024) -> fp32 CHW tile on the zero 1024 x 1024
// canvas, top-left aligned: ToTensor (/255), Normalize(ImageNet mean/std) (dataloader_coco.py:286-292,
// augmentation.py:229-249) and the fixed-1024 zero padding of nested_tensor_from_tensor_list
// (utils/misc.py:46-67) in one pass.  in [B,h,w,3] u8, out [B,3,1024,1024] fp32.  One thread per 4 pixels.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                            int B, int h, int w) {
#pragma clang fp contract(off)
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int64_t total = (int64_t)B * 1024 * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x4 = (int)(i & 255) * 4;
        const int y = (int)((i >> 8) & 1023);
        const int64_t b = i >> 18;
        f32x4 v[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (y < h) {
            const unsigned char* row = in + ((b * h + y) * (int64_t)w) * 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x4 + j;
                if (x < w) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c][j] = ((float)row[x * 3 + c] / 255.0f - mean[c]) / stdv[c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) *(f32x4*)(out + ((b * 3 + c) * 1024 + y) * (int64_t)1024 + x4) = v[c];
    }
}

// ---------------------------------------------------------------------------
// Input pipeline with the val transform's resize (dataloader_coco.py:286-292 -> augmentation.py:77-133
// -> torchvision F.resize on a PIL image = PIL bilinear resample).  The arithmetic is Pillow's 8-bit ImagingResample:
// separable antialiased triangle filter, 22-bit fixed-point coefficients (computed on the host in double exactly as
// Resample.c does, wm_api.hip: resize_coeffs), horizontal pass into an 8-bit image, then the vertical pass, each output
// clip8((sum + 2^21) >> 22) -- integer work, bit-exact with PIL.  The second kernel fuses the vertical pass with ToTensor,
// Normalize and the zero padding to 1024 x 1024 (utils/misc.py:46-67).
// ---------------------------------------------------------------------------
constexpr int RESIZE_PREC_BITS = 22;

// in [B,h,w,3] u8 -> tmp [B,h,ow,3] u8; bounds [ow][2] = (first input column, taps), kk [ow][ksize]
__global__ __launch_bounds__(256) void resize_h_u8_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ tmp,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                          int B, int h, int w, int ow) {
    const int64_t total = (int64_t)B * h * ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xx = (int)(i % ow);
        const int64_t by = i / ow;                                   // b * h + y
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        const unsigned char* src = in + (by * w + x0) * 3;
        const int* k = kk + (int64_t)xx * ksize;
        int a0 = 1 << (RESIZE_PREC_BITS - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < n; ++x) {
            const int c = k[x];
            a0 += src[3 * x] * c; a1 += src[3 * x + 1] * c; a2 += src[3 * x + 2] * c;
        }
        unsigned char* dst = tmp + i * 3;
        dst[0] = (unsigned char)min(max(a0 >> RESIZE_PREC_BITS, 0), 255);
        dst[1] = (unsigned char)min(max(a1 >> RESIZE_PREC_BITS, 0), 255);
        dst[2] = (unsigned char)min(max(a2 >> RESIZE_PREC_BITS, 0), 255);
    }
}

// tmp [B,h,ow,3] u8 -> out [B,3,1024,1024] fp32: vertical pass to oh rows, /255, ImageNet normalise, zero canvas outside
__global__ __launch_bounds__(256) void resize_v_normalize_kernel(const unsigned char* __restrict__ tmp, float* __restrict__ out,
                                                                 const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                                 int B, int h, int ow, int oh) {
#pragma clang fp contract(off)
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int64_t total = (int64_t)B * 1024 * 1024;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xx = (int)(i & 1023), yy = (int)((i >> 10) & 1023);
        const int64_t b = i >> 20;
        float v[3] = {0.f, 0.f, 0.f};
        if (yy < oh && xx < ow) {
            const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
            const unsigned char* src = tmp + ((b * h + y0) * (int64_t)ow + xx) * 3;
            const int* k = kk + (int64_t)yy * ksize;
            int a[3] = {1 << (RESIZE_PREC_BITS - 1), 1 << (RESIZE_PREC_BITS - 1), 1 << (RESIZE_PREC_BITS - 1)};
            for (int y = 0; y < n; ++y) {
                const int c = k[y];
                const unsigned char* p = src + (int64_t)y * ow * 3;
                a[0] += p[0] * c; a[1] += p[1] * c; a[2] += p[2] * c;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int u = min(max(a[c] >> RESIZE_PREC_BITS, 0), 255);
                v[c] = ((float)u / 255.0f - mean[c]) / stdv[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((b * 3 + c) * 1024 + yy) * (int64_t)1024 + xx] = v[c];
    }
}

// ---- round 4: the two resize passes at streaming rate (the generic kernels above: one thread per output pixel, byte loads from
// global memory, 110 us per 3648 x 5472 frame = 0.08 of the HBM rate for 72 MB of algorithmic traffic) -------------------------
// Horizontal pass, row-staged: a workgroup walks input rows; a row (w * 3 bytes) is staged in LDS by coalesced dword loads, each
// thread owns up to OPT output columns and keeps their taps' coefficients in registers for all its rows (KMAX taps, zero beyond
// the column's count: a zero coefficient times any staged byte adds nothing, and the LDS row has KMAX * 3 bytes of slack).
// Same integer arithmetic as resize_h_u8_kernel: bit-identical.
template <int KMAX, int OPT>
__global__ __launch_bounds__(256) void resize_h_rows_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ tmp,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                            int64_t rows_total, int w, int ow, int rows_per_block, int64_t in_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned srow[];
    const int tid = threadIdx.x;
    int x0[OPT], coef[OPT][KMAX];
    bool valid[OPT];
#pragma unroll
    for (int o = 0; o < OPT; ++o) {
        const int xx = tid + 256 * o;
        valid[o] = xx < ow;
        const int n = valid[o] ? bounds[2 * xx + 1] : 0;
        x0[o] = valid[o] ? bounds[2 * xx] : 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) coef[o][k] = k < n ? kk[(int64_t)xx * ksize + k] : 0;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows_total ? r0 + rows_per_block : rows_total;
    const int row_bytes = w * 3;
    for (int64_t row = r0; row < r1; ++row) {
        const int64_t byte0 = row * row_bytes, a0 = byte0 & ~(int64_t)3;
        const int shift = (int)(byte0 - a0), ndw = (shift + row_bytes + 3) >> 2;
        for (int i = tid; i < ndw; i += 256) {
            const int64_t off = a0 + 4 * (int64_t)i;
            unsigned v;
            if (off + 4 <= in_bytes) v = *(const unsigned*)(in + off);
            else {                                            // the last dword of the whole buffer: byte by byte
                v = 0;
                for (int j = 0; j < 4; ++j)
                    if (off + j < in_bytes) v |= (unsigned)in[off + j] << (8 * j);
            }
            srow[i] = v;
        }
        __syncthreads();
        const unsigned char* sb = (const unsigned char*)srow + shift;
#pragma unroll
        for (int o = 0; o < OPT; ++o) {
            if (!valid[o]) continue;
            const unsigned char* src = sb + x0[o] * 3;
            int a0c = 1 << (RESIZE_PREC_BITS - 1), a1c = a0c, a2c = a0c;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int c = coef[o][k];
                a0c += src[3 * k] * c; a1c += src[3 * k + 1] * c; a2c += src[3 * k + 2] * c;
            }
            unsigned char* dst = tmp + (row * ow + tid + 256 * o) * 3;
            dst[0] = (unsigned char)min(max(a0c >> RESIZE_PREC_BITS, 0), 255);
            dst[1] = (unsigned char)min(max(a1c >> RESIZE_PREC_BITS, 0), 255);
            dst[2] = (unsigned char)min(max(a2c >> RESIZE_PREC_BITS, 0), 255);
        }
        __syncthreads();
    }
}

// Vertical pass + ToTensor + Normalize + zero canvas, four output pixels per thread: a workgroup is one output row (its taps'
// coefficients are wave-uniform), a thread reads 12 contiguous bytes (3 dwords) per tap row and writes one 16-byte chunk per
// channel.  ow % 4 == 0.  Same arithmetic as resize_v_normalize_kernel: bit-identical.
__global__ __launch_bounds__(256) void resize_v_normalize4_kernel(const unsigned char* __restrict__ tmp, float* __restrict__ out,
                                                                  const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                                  int h, int ow, int oh) {
#pragma clang fp contract(off)
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int yy = blockIdx.x & 1023;
    const int64_t b = blockIdx.x >> 10;
    const int xx4 = threadIdx.x * 4;
    f32x4 v[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (yy < oh && xx4 < ow) {
        const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
        const unsigned* src = (const unsigned*)(tmp + ((b * h + y0) * (int64_t)ow + xx4) * 3);
        const int* k = kk + (int64_t)yy * ksize;
        int a[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) a[j] = 1 << (RESIZE_PREC_BITS - 1);
        const int64_t stride_dw = (int64_t)ow * 3 / 4;
        for (int y = 0; y < n; ++y) {
            const int c = k[y];
            const unsigned d0 = src[0], d1 = src[1], d2 = src[2];
            src += stride_dw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] += (int)((d0 >> (8 * j)) & 255u) * c;
                a[4 + j] += (int)((d1 >> (8 * j)) & 255u) * c;
                a[8 + j] += (int)((d2 >> (8 * j)) & 255u) * c;
            }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int u = min(max(a[px * 3 + c] >> RESIZE_PREC_BITS, 0), 255);
                v[c][px] = ((float)u / 255.0f - mean[c]) / stdv[c];
            }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) *(f32x4*)(out + ((b * 3 + c) * 1024 + yy) * (int64_t)1024 + xx4) = v[c];
}

}  // namespace wm
