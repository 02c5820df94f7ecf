// Flash-style multi-head attention on MFMA 32x32x16 (bf16 / fp16), gfx950.
//
// Two kernels share the per-wave core below:
//   attn_global_kernel  - 4096 (or any multiple of 64) keys per head, optional
//                         decomposed rel-pos bias: the 4 global blocks
//                         (image_encoder.py:246-262, 347-383) and, without the
//                         bias, the HFC cross-attention (image_encoder.py:500-503).
//   attn_window_kernel  - 14x14 windows with zero-padded tokens that still act
//                         as keys/values (image_encoder.py:190-199, 265-311).
//
// Per wave: 32 query rows, scores computed TRANSPOSED (S^T = K Q^T) so that a
// lane owns one query column: its 32x32 accumulator registers are that query's
// scores for 16 of the tile's 32 keys, the partner lane (lane^32) holds the other
// 16.  Softmax is therefore lane-local plus one cross-half exchange, and the
// exponentiated tile is already the B operand of the P*V product
// (O^T = V^T P^T, cdna_hip_programming.md §3 "An accumulator tile as the next
// MFMA's operand"), whose A operand V^T comes from the row-major V tile in LDS
// through ds_read_b64_tr_b16 (T10).
//
// The rel-pos bias is never materialised per (query,key) pair in memory:
//   bias[q,(kh,kw)] = q.Rh[qh-kh+S-1] + q.Rw[qw-kw+S-1]   (unscaled q, :376-381)
// For global attention a key tile is one grid row (kh fixed, kw = 0..63), so the
// kw-term is the same 64-vector for every tile (kept in registers, used as the
// MFMA accumulator's initial value) and the kh-term is one scalar per tile.
// Both are produced in the prologue by MFMA products Q x table^T.
#pragma once
#include "wm_common.h"

#ifndef WM_DEV_TIMELINE
#define WM_DEV_TIMELINE 0
#endif

namespace wm {

struct AttnArgs {
    const u16* q; const u16* k; const u16* v;   // 16-bit, row = token
    u16* out;
    int q_stride, k_stride, v_stride, out_stride;   // elements between consecutive tokens
    int nq, nk;                                     // tokens per image (queries / keys)
    float scale;                                    // head_dim^-0.5
    const float* rel_h; const float* rel_w;         // [2*S-1, HD] fp32 or null
    const float* qkv_bias;                          // window kernel: [3*D] fp32 (padded tokens)
    const u16* qkv_bias16;                          // the same, rounded to the operand type: a padded token's K / V row
    int heads;
    // q carries scale * log2(e) ("Scores" below): the engine folds it into the q rows of the qkv weight (one rounding, as before); the
    // single-op entry points scale a copy of q first (scale_q16_kernel, one more rounding)
    unsigned char* out8;                            // WM_PREC_FP8: write the output as e4m3 bytes (row stride out_stride bytes) instead of 16-bit
#if WM_DEV_TIMELINE
    unsigned long long* tl;                         // dev build: s_memtime stamps of workgroup 0 ([wave][64]) or null
#endif
};

template <int HD> struct AttnGeom {
    static constexpr int KS = HD * 2 + 16;                      // K row stride (bytes): odd multiple of 16 B
    static constexpr int VS = (HD == 128) ? 320 : 192;          // V row stride (bytes): odd multiple of 64 B
    static constexpr int NKS = HD / 16;                         // QK^T k-steps
    static constexpr int NDT = (HD + 31) / 32;                  // 32-row O^T tiles
    static constexpr int CH = HD / 8;                           // 16-byte chunks per row
    // HD = 80: the last 32-row O^T tile has 16 spare rows.  The V image's pad column HD is set to 1.0 once, so row HD
    // of O^T = sum_k P[k][q] = the softmax denominator, from the matrix pipe instead of 32 v_add per tile (and it is
    // the sum of exactly the rounded P values the numerator uses).  Lane (c, h = 0) holds it in o[NDT-1][LSUM_R].
    static constexpr bool LSUM_IN_O = (HD % 32) != 0;
    static constexpr int LSUM_R = ((HD % 32) / 8) * 4;
    static_assert(!LSUM_IN_O || (HD % 8) == 0, "pad column must fall on accumulator register LSUM_R of half 0");
};

template <class T>
__device__ __forceinline__ typename T::vec8 lds_read_v8(const char* p) {
    return *(const typename T::vec8*)p;
}

// V^T fragment for one 32x32x16 k-step: two transposed reads of 4 keys x 16 dims.
template <class T>
__device__ __forceinline__ typename T::vec8 lds_read_vT(const char* p_first, int second_off) {
    typedef __attribute__((address_space(3))) s16x4* lptr;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p_first));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p_first + second_off));
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return __builtin_bit_cast(typename T::vec8, r);
}

// Online-softmax state of one wave (32 queries, lane = query column + 32*half).
template <int NDT> struct SoftmaxState {
    float m;          // reference point of the exponentials (log2 domain); scores reach the softmax RELATIVE to it (see "Scores" below)
    float l;          // running sum, this lane's half of the keys only
    f32x16 o[NDT];    // O^T accumulators
    __device__ __forceinline__ void init() {
        m = 0.f; l = 0.f;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
    }
};

// Scores (round 4).  Q reaches the kernels multiplied by c1 = softmax scale * log2 e (folded into the q rows of the qkv weight
// before its one rounding; AttnArgs::q_prescaled), so the QK^T accumulators ARE the log2-domain scores, and everything that used to
// be added per score on the vector pipe rides the matrix pipe instead:
//   - the kw rel-pos term: the accumulators' initial value (as before);
//   - the per-(query, key tile) scalars -- the kh rel-pos term and minus the reference point m -- through ONE extra 16-deep k-step
//     of the QK^T product: B[k][query] holds the scalar as a (hi, lo) pair of 16-bit values (22 / 16 significant bits), A[key][k] is
//     1.0 at that pair's two k positions and 0 elsewhere, the same for every key of the tile (bias_a_frag / bias_b_*).  A B
//     fragment carries the pairs of 8 consecutive tiles (4 per lane half) and is rebuilt every 8 tiles and when m moves.
// The softmax is then max (the deferred-rescale check), exp2 of the accumulator itself, convert: the FMA per score is gone
// (32 of ~116 vector instructions per 64-key tile in the global kernel).  m moves only when some query's maximum exceeds it by more
// than RESCALE_THR (log2 units) -- and at the first tile, where it becomes that tile's maximum -- so P <= 2^RESCALE_THR: harmless in
// fp32 accumulators and for 16-bit floating P.  When it moves, the tile's scores are corrected on the vector pipe (rare).
constexpr float RESCALE_THR = 6.0f;

template <class T> __device__ __forceinline__ unsigned one_pair_bits() {
    return std::is_same<T, FP16>::value ? 0x3C003C00u : 0x3F803F80u;           // (1.0, 1.0) as two 16-bit floats
}
// A fragment of the bias k-step for a lane holding k = 8 h .. 8 h + 7: 1.0 at k = 2 pos, 2 pos + 1 if `mine`, else 0
template <class T>
__device__ __forceinline__ typename T::vec8 bias_a_frag(int pos, bool mine) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 a;
#pragma unroll
    for (int d = 0; d < 4; ++d) a[d] = (mine && d == pos) ? one_pair_bits<T>() : 0u;
    return __builtin_bit_cast(typename T::vec8, a);
}
template <class T>
__device__ __forceinline__ void hi_lo(float v, typename T::elem& hi, typename T::elem& lo) {
    hi = T::from_f32(v);
    lo = T::from_f32(v - T::to_f32(hi));
}
// B fragment, no per-tile term: (hi, lo) of `v` at k = 0, 1 (lanes of half 0; bias_a_frag(0, h == 0) selects them)
template <class T>
__device__ __forceinline__ typename T::vec8 bias_b_const(float v) {
    typename T::vec8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = T::from_f32(0.f);
    typename T::elem hi, lo;
    hi_lo<T>(v, hi, lo);
    b[0] = hi; b[1] = lo;
    return b;
}
// B fragment with the kh rel-pos term: lane (c, h) holds the pairs of tiles (j & ~7) + 4 h + i, i = 0..3, each rel_h[tile][c] - m
// (tile j is selected by bias_a_frag(j & 3, h == ((j >> 2) & 1))); rel_h: [tile][32 queries] fp32 in LDS
template <class T>
__device__ __forceinline__ typename T::vec8 bias_b_rel(const float* rel_h, int j, int c, int h, float m, int ntiles) {
    typename T::vec8 b;
    const int j0 = (j & ~7) + 4 * h;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = rel_h[min(j0 + i, ntiles - 1) * 32 + c];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        typename T::elem hi, lo;
        hi_lo<T>(v[i] - m, hi, lo);
        b[2 * i] = hi; b[2 * i + 1] = lo;
    }
    return b;
}
// q -> c1 q for callers that hold the reference's plain q (the single-op entry points): out[row][0..cols) = round16(c1 * in[row][0..cols))
template <class T>
__global__ __launch_bounds__(256) void scale_q16_kernel(const u16* __restrict__ in, int in_stride, u16* __restrict__ out, int64_t rows, int cols, float c1) {
    const int cpr = cols / 8;
    const int64_t n = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr);
        typename T::vec8 v = *(const typename T::vec8*)(in + row * in_stride + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = T::from_f32(T::to_f32(v[j]) * c1);
        *(typename T::vec8*)(out + row * cols + ch * 8) = v;
    }
}

// One key tile of NT*32 keys (4-wave kernel): s[t] hold the log2-domain scores without the tile's scalar `tile_bias` (the kh rel-pos
// term, 0 without rel-pos) and without the reference point: both are added per score here (one v_add in front of the exp2).  The
// 8-wave kernel's non-rel-pos instances move that add to the matrix pipe (bias k-step, "Scores"); for this kernel's shapes it measured
// slower (head_dim 128: 1257 vs 1190 us, 34 MFMAs per tile against a vector phase that is already the shorter one).  sV: the tile's V rows.
template <class T, int HD, int NT>
__device__ __forceinline__ void softmax_pv(SoftmaxState<AttnGeom<HD>::NDT>& st, f32x16 (&s)[NT], bool first, float tile_bias, const char* sV, int lane) {
    using G = AttnGeom<HD>;
    float mx0 = -1e30f, mx1 = -1e30f;                       // two chains: a dependent v_max3 issues every ~8 cycles, not 4
#pragma unroll
    for (int r = 0; r < 16; ++r) { mx0 = fmaxf(mx0, s[0][r]); mx1 = fmaxf(mx1, s[NT - 1][r]); }
    float mx = fmaxf(mx0, mx1) + (tile_bias - st.m);        // this tile's maximum relative to the reference point (same association as attn_glob8.h)
    {   // the other half of the keys sits in lane ^ 32
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (first || !__all(mx <= RESCALE_THR)) {
        const float d = first ? mx : fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        st.l *= alpha;
#pragma unroll
        for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[dt][r] *= alpha;
        st.m += d;
    }
    const float off = tile_bias - st.m;
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[t][r] + off);
            s[t][r] = pv;
            if constexpr (!G::LSUM_IN_O) ls += pv;      // else: row HD of O^T accumulates the sum (V pad column = 1)
        }
    st.l += ls;

    // P^T fragments -> O^T += V^T P^T.  k-step ks covers keys 16*ks .. 16*ks+15 of the tile.
    const int g = lane >> 4;
    // transposed-read address: group g = 16-lane group; half h = g>>1 picks keys +4h, (g&1) picks dims +16
    const int lq = (lane & 15) >> 2, lp = lane & 3;
    const int v_lane_off = (4 * (g >> 1) + lq) * G::VS + (16 * (g & 1) + 4 * lp) * 2;
#pragma unroll
    for (int ks = 0; ks < 2 * NT; ++ks) {
        typename T::vec8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = T::from_f32_bounded(s[ks >> 1][8 * (ks & 1) + j]);     // P <= 2^RESCALE_THR
#pragma unroll
        for (int dt = 0; dt < G::NDT; ++dt) {
            const char* p = sV + (16 * ks) * G::VS + dt * 64 + v_lane_off;
            typename T::vec8 va = lds_read_vT<T>(p, 8 * G::VS);
            st.o[dt] = T::mfma32(va, pb, st.o[dt]);
        }
    }
}

// S^T[t] += K[tile rows 32t..32t+31] * Q^T over HD, K rows in LDS with stride KS.
template <class T, int HD, int NT>
__device__ __forceinline__ void qk_tile(f32x16 (&s)[NT], const typename T::vec8 (&qf)[AttnGeom<HD>::NKS],
                                        const char* sK, int lane) {
    using G = AttnGeom<HD>;
    const int r31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            typename T::vec8 kf = lds_read_v8<T>(sK + (32 * t + r31) * G::KS + (16 * ks + 8 * h) * 2);
            s[t] = T::mfma32(kf, qf[ks], s[t]);
        }
}

// Set the pad column HD of `rows` V rows (stride VS) to 1.0 (see AttnGeom::LSUM_IN_O).
template <class T, int HD>
__device__ __forceinline__ void v_pad_ones(char* sV, int rows, int tid, int nthreads) {
    using G = AttnGeom<HD>;
    if constexpr (G::LSUM_IN_O) {
        const typename T::elem one = T::from_f32(1.0f);
        for (int r = tid; r < rows; r += nthreads) *(typename T::elem*)(sV + r * G::VS + HD * 2) = one;
    }
}

// Normalise and store O^T: lane (c = lane&31, h) holds dims 32dt + (r&3) + 8(r>>2) + 4h of query c.
template <class T, int HD>
__device__ __forceinline__ void store_out(SoftmaxState<AttnGeom<HD>::NDT>& st, u16* out_row, int lane, bool valid, unsigned char* out8_row = nullptr) {
    using G = AttnGeom<HD>;
    const int h = lane >> 5;
    float l;
    if constexpr (G::LSUM_IN_O) l = __shfl(st.o[G::NDT - 1][G::LSUM_R], lane & 31, 64);
    else l = st.l + __shfl_xor(st.l, 32, 64);
    const float inv = 1.0f / l;
    if (!valid) return;
#pragma unroll
    for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int d = 32 * dt + 8 * rg + 4 * h;
            if (d < HD) {
                if (out8_row) {                              // wave-uniform: e4m3 A operand of the fp8 proj GEMM (gemm8.h)
                    const f32x4 v{st.o[dt][4 * rg] * inv, st.o[dt][4 * rg + 1] * inv, st.o[dt][4 * rg + 2] * inv, st.o[dt][4 * rg + 3] * inv};
                    *(unsigned*)(out8_row + d) = pack4_e4m3(v);
                } else {
                    typename T::vec4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = T::from_f32_bounded(st.o[dt][4 * rg + j] * inv);     // a convex combination of V rows
                    *(typename T::vec4*)(out_row + d) = o;
                }
            }
        }
}

// The same through a per-wave LDS image [32 queries][ROW_STRIDE bytes]: normalised 16-bit rows are written as the accumulators hold
// them (8 B per lane, one query per lane) and leave as 16-B chunks of whole rows, `row_ptr(r)` giving query r's output row or null.
// A row-per-lane store instruction touches 32 different 128-B lines (20 such stores per item: ~3.4k cycles of an 18k-cycle window
// item in the timeline); a chunked one touches ~8.
template <class T, int HD, class RowPtr>
__device__ __forceinline__ void store_out_rows(SoftmaxState<AttnGeom<HD>::NDT>& st, char* stage, int lane, RowPtr row_ptr) {
    using G = AttnGeom<HD>;
    constexpr int RS = HD * 2 + 16;
    const int c = lane & 31, h = lane >> 5;
    float l;
    if constexpr (G::LSUM_IN_O) l = __shfl(st.o[G::NDT - 1][G::LSUM_R], c, 64);
    else l = st.l + __shfl_xor(st.l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int d = 32 * dt + 8 * rg + 4 * h;
            if (d < HD) {
                typename T::vec4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = T::from_f32_bounded(st.o[dt][4 * rg + j] * inv);     // a convex combination of V rows
                *(typename T::vec4*)(stage + c * RS + d * 2) = o;
            }
        }
    constexpr int CH = HD / 8, NCHUNK = 32 * CH;
#pragma unroll
    for (int i = 0; i < (NCHUNK + 63) / 64; ++i) {
        const int e = lane + 64 * i;
        if (e < NCHUNK) {
            const int r = e / CH, ch = e % CH;
            u16* dst = row_ptr(r);
            if (dst) *(s16x8*)(dst + ch * 8) = *(const s16x8*)(stage + r * RS + ch * 16);
        }
    }
}

// ---------------------------------------------------------------------------
// Global attention (key tile = 64 keys = one grid row when REL)
// grid (nq/128, heads, batch), 256 threads.
// ---------------------------------------------------------------------------
template <int HD, bool REL> struct GlobalLds {
    using G = AttnGeom<HD>;
    static constexpr int WAVE_F = 32 * 65;                                    // floats per wave (padded staging)
    static constexpr int RELH_BYTES = REL ? 4 * WAVE_F * 4 : 0;              // [wave][kh][query] fp32, aliased with [query][65] staging
    static constexpr int K_BYTES = 64 * G::KS, V_BYTES = 64 * G::VS;
    static constexpr int KV_OFF = RELH_BYTES;
    static constexpr int TOTAL = RELH_BYTES + 2 * (K_BYTES + V_BYTES);
};

template <class T, int HD, bool REL>
__global__ __launch_bounds__(256, 2) void attn_global_kernel(AttnArgs p) {
    using G = AttnGeom<HD>;
    using L = GlobalLds<HD, REL>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware: the workgroups of one (image, head) share K / V and get one XCD's L2 (attn_glob8.h has the measurement)
    const int nqb = p.nq / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = (lid / nqb) % p.heads, b = lid / (nqb * p.heads);
    const int q0 = (lid % nqb) * 128 + wave * 32;

    const u16* qb = p.q + ((size_t)b * p.nq) * p.q_stride + head * HD;
    const u16* kb = p.k + ((size_t)b * p.nk) * p.k_stride + head * HD;
    const u16* vb = p.v + ((size_t)b * p.nk) * p.v_stride + head * HD;

    // Q fragments (B operand): lane holds Q[q0+c][16ks + 8h .. +7], in the log2 domain (see "Scores")
    typename T::vec8 qf[G::NKS];
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks)
        qf[ks] = *(const typename T::vec8*)(qb + (size_t)(q0 + c) * p.q_stride + 16 * ks + 8 * h);

    char* sKV = smem + L::KV_OFF;
    f32x16 relw[2];
    float* sRelH = (float*)smem + wave * (32 * 65);

    if constexpr (REL) {
        // ---- prologue: rel_w (registers) and rel_h (LDS) for this wave's 32 queries ----
        // table image: 128 rows x HD 16-bit, row stride KS, rows >= 127 zero
        const int qh = q0 >> 6, qw0 = q0 & 63;
        const float inv_scale = 1.0f / p.scale;
        char* sTab = sKV;
        float* sT = (float*)smem + wave * (32 * 65);          // [query c][65] fp32 staging (padded: conflict-free)
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const float* tab = which == 0 ? p.rel_w : p.rel_h;
            __syncthreads();
            for (int e = tid; e < 128 * (HD / 4); e += 256) {
                const int row = e / (HD / 4), c4 = e % (HD / 4);
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row < 127) v = *(const f32x4*)(tab + (size_t)row * HD + c4 * 4);
                typename T::vec4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                *(typename T::vec4*)(sTab + row * G::KS + c4 * 8) = o;
            }
            __syncthreads();
            if (which == 0) {
                // T_w^T[i][c] = Rw[i].q_c for i in two passes of 64; pick i = qw + 63 - kw
                const int qw = qw0 + c;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) relw[t][r] = 0.f;
#pragma unroll 1
                for (int pass = 0; pass < 2; ++pass) {
                    f32x16 acc[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                    qk_tile<T, HD, 2>(acc, qf, sTab + pass * 64 * G::KS, lane);
                    // stage as [c][i_local]
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int il = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                            sT[c * 65 + il] = acc[t][r];
                        }
                    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS RAW
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int kw = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                            const int idx = qw + 63 - kw;
                            if ((idx >> 6) == pass) relw[t][r] = sT[c * 65 + (idx & 63)] * inv_scale;
                        }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                }
            } else {
                // T_h^T[kh][c] = Rh[qh + 63 - kh].q_c : A rows taken in reversed order
                f32x16 acc[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                const int r31 = lane & 31;
#pragma unroll
                for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = qh + 63 - (32 * t + r31);
                        typename T::vec8 kf = lds_read_v8<T>(sTab + row * G::KS + (16 * ks + 8 * h) * 2);
                        acc[t] = T::mfma32(kf, qf[ks], acc[t]);
                    }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kh = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                        sRelH[kh * 32 + c] = acc[t][r] * inv_scale;
                    }
            }
        }
        __syncthreads();
    }

    // ---- main loop over key tiles of 64 ----
    const int ntiles = p.nk / 64;
    constexpr int NCH = 64 * G::CH;                 // 16-B chunks per K (or V) tile
    constexpr int PER = (NCH + 255) / 256;
    s16x8 kreg[PER], vreg[PER];

    // per-thread source pointers of the staging chunks, advanced by one tile per issue (no per-tile address math)
    const u16* kp[PER];
    const u16* vp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = min(tid + i * 256, NCH - 1);
        kp[i] = kb + (size_t)(e / G::CH) * p.k_stride + (e % G::CH) * 8;
        vp[i] = vb + (size_t)(e / G::CH) * p.v_stride + (e % G::CH) * 8;
    }
    const size_t k_step = (size_t)64 * p.k_stride, v_step = (size_t)64 * p.v_stride;
    auto issue = [&](int) {                          // tiles are requested in order
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (tid + i * 256 < NCH) {
                kreg[i] = *(const s16x8*)kp[i];
                vreg[i] = *(const s16x8*)vp[i];
            }
            kp[i] += k_step;
            vp[i] += v_step;
        }
    };
    v_pad_ones<T, HD>(sKV + L::K_BYTES, 64, tid, 256);
    v_pad_ones<T, HD>(sKV + (L::K_BYTES + L::V_BYTES) + L::K_BYTES, 64, tid, 256);
    auto commit = [&](int buf) {
        char* sK = sKV + buf * (L::K_BYTES + L::V_BYTES);
        char* sV = sK + L::K_BYTES;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * 256;
            if (e < NCH) {
                const int key = e / G::CH, ch = e % G::CH;
                *(s16x8*)(sK + key * G::KS + ch * 16) = kreg[i];
                *(s16x8*)(sV + key * G::VS + ch * 16) = vreg[i];
            }
        }
    };

    SoftmaxState<G::NDT> st;
    st.init();
    issue(0);
    commit(0);
    __syncthreads();

    for (int j = 0; j < ntiles; ++j) {
        const int buf = j & 1;
        if (j + 1 < ntiles) issue(j + 1);
        const char* sK = sKV + buf * (L::K_BYTES + L::V_BYTES);
        const char* sV = sK + L::K_BYTES;
        f32x16 s[2];
        float rh = 0.f;
        if constexpr (REL) {
            rh = sRelH[j * 32 + c];                       // kh-term: one scalar per query and tile, added in front of the exp2
#pragma unroll
            for (int t = 0; t < 2; ++t) s[t] = relw[t];    // kw-term: the accumulators' initial value
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        }
        qk_tile<T, HD, 2>(s, qf, sK, lane);
        softmax_pv<T, HD, 2>(st, s, j == 0, rh, sV, lane);
        if (j + 1 < ntiles) commit(buf ^ 1);
        __syncthreads();
    }
    u16* orow = p.out + ((size_t)b * p.nq + q0 + c) * p.out_stride + head * HD;
    unsigned char* orow8 = p.out8 ? p.out8 + ((size_t)b * p.nq + q0 + c) * p.out_stride + head * HD : nullptr;
    store_out<T, HD>(st, orow, lane, true, orow8);
}

// ---------------------------------------------------------------------------
// Window attention: one workgroup per (tile, window, head); all 196 keys of the
// window (incl. padded tokens, whose k/v are the qkv bias) resident in LDS.
// grid (25, heads, batch), 448 threads = 7 waves, wave w owns query slots 32w..32w+31.
//
// Rel-pos bias per wave: T[c][i] = q_c . table[i] by one MFMA pass over the 64-row
// table image (rel_h rows 0..26, rel_w rows 32..58), staged per wave in LDS; each
// lane then gathers its query's 14 + 14 values U[kh] = T[qh-kh+13], V[kw] = T[32+qw-kw+13]
// into registers.  Key slots are laid out 14 x 16 (two zero pad columns), so that in the
// fully unrolled key loop a score's bias is one add of two registers (see the key loop).
// ---------------------------------------------------------------------------
template <int HD> struct WindowLds {
    using G = AttnGeom<HD>;
    static constexpr int NKEY = 224;                                       // 196 padded to 7 x 32
    static constexpr int NWAVE = 7;
    static constexpr int K_BYTES = NKEY * G::KS, V_BYTES = NKEY * G::VS;
    static constexpr int TAB_BYTES = 64 * G::KS;                           // rel_h rows 0..26, rel_w rows 32..58
    static constexpr int T_BYTES = NWAVE * 32 * 65 * 4;                    // per wave [query][65] fp32
    static constexpr int K_OFF = 0, V_OFF = K_BYTES, TAB_OFF = V_OFF + V_BYTES, T_OFF = TAB_OFF + TAB_BYTES;
    static constexpr int TOTAL = T_OFF + T_BYTES;
};

template <class T, int HD>
__global__ __launch_bounds__(448, 2) void attn_window_kernel(AttnArgs p, int nitems) {
    using G = AttnGeom<HD>;
    using L = WindowLds<HD>;
    constexpr int WS = 14, GRID = 64, NWIN = 5, NTOK = WS * WS, NTHR = 448;
    constexpr int NPF = (L::NKEY * G::CH) / NTHR;                // 16-byte K (and V) chunks per thread per item
    static_assert((L::NKEY * G::CH) % NTHR == 0, "staging split");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int D = p.heads * HD;
    const float inv_scale = 1.0f / p.scale;

    char* sK = smem + L::K_OFF;
    char* sV = smem + L::V_OFF;
    char* sTab = smem + L::TAB_OFF;
    float* sT = (float*)(smem + L::T_OFF) + wave * (32 * 65);

    // Persistent: one workgroup per CU walks items (tile, window, head).  All 196 keys of a window live in
    // LDS (one workgroup per CU), so nothing else on the CU could hide the latency of staging them: the next
    // item's K / V chunks and Q fragments are fetched into registers while the current item computes.
    // Item order: heads of one window are neighbours and, through the XCD remap, share an L2.
    auto decode = [&](int item, int& b, int& win, int& head) {
        head = item % p.heads;
        win = (item / p.heads) % (NWIN * NWIN);
        b = item / (p.heads * NWIN * NWIN);
    };

    s16x8 kreg[NPF], vreg[NPF];
    auto prefetch_kv = [&](int item) {
        int b, win, head;
        decode(item, b, win, head);
        const int wy = win / NWIN, wx = win % NWIN;
        const u16* kbase = p.k + ((size_t)b * GRID * GRID) * p.k_stride + head * HD;
        const u16* vbase = p.v + ((size_t)b * GRID * GRID) * p.v_stride + head * HD;
        // the chunk coordinates are re-derived per item from an opaque copy of tid: hoisted out of the item loop they were spilled, and
        // every reload (scratch = vector memory) came with a vmcnt(0) that drained the prefetch loads issued before it
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid_o + i * NTHR;
            const int key = e / G::CH, ch = e % G::CH;                                  // key slot = 16 kh + kw (kw 14, 15: zero rows)
            // A token outside the image is zero after norm1, so its qkv row is the bias (image_encoder.py:190-194, 281); the two pad
            // columns of the 14 x 16 slot layout take the bias row too: their scores carry the -1e30 column bias, so P = 0 exactly
            // whatever finite K / V they hold.  The row pointer is SELECTED bitwise -- as `if`s this was two exec-mask branches per
            // chunk, and converting the fp32 bias here put 4 loads and a vmcnt(0) in the middle of every edge window's prefetch
            // (timeline: 3-9k of an item's 23k cycles went into issuing it).
            const int y = wy * WS + (key >> 4), x = wx * WS + (key & 15);
            const bool in = (key & 15) < WS && y < GRID && x < GRID;
            const size_t tok = (size_t)(min(y, GRID - 1) * GRID + min(x, GRID - 1));
            const size_t msk = (size_t)0 - (size_t)in;
            const u16* krow = (const u16*)(((size_t)(kbase + tok * p.k_stride) & msk) | ((size_t)(p.qkv_bias16 + D + head * HD) & ~msk));
            const u16* vrow = (const u16*)(((size_t)(vbase + tok * p.v_stride) & msk) | ((size_t)(p.qkv_bias16 + 2 * D + head * HD) & ~msk));
            const s16x8 kv8 = *(const s16x8*)(krow + ch * 8);
            const s16x8 vv8 = *(const s16x8*)(vrow + ch * 8);
            kreg[i] = kv8;
            vreg[i] = vv8;
        }
    };
    auto commit_kv = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + i * NTHR;
            const int key = e / G::CH, ch = e % G::CH;
            *(s16x8*)(sK + key * G::KS + ch * 16) = kreg[i];
            *(s16x8*)(sV + key * G::VS + ch * 16) = vreg[i];
        }
    };
    // this wave's 32 query slots of an item: validity, token, Q fragments
    const int qi = wave * 32 + c;                         // slot in the window (0..223)
    const int qh = qi / WS, qw = qi - qh * WS;
    struct QInfo { bool valid; size_t row; };
    auto q_info = [&](int item) {
        int b, win, head;
        decode(item, b, win, head);
        const int y = (win / NWIN) * WS + qh, x = (win % NWIN) * WS + qw;
        const bool valid = (qi < NTOK) && (y < GRID) && (x < GRID);
        const size_t tok = valid ? (size_t)(y * GRID + x) : 0;
        return QInfo{valid, (size_t)b * GRID * GRID + tok};
    };
    // (Fetching Q as 16-B chunks of whole rows -- ~14 lines per load instruction instead of 32 -- and forming the fragments through
    // LDS was tried: the five divisions per lane and the LDS round trip cost more than the lines saved, +4 % per launch.)
    auto load_q = [&](typename T::vec8 (&qf)[G::NKS], int item) {
        int b, win, head;
        decode(item, b, win, head);
        const QInfo qi_ = q_info(item);
        const u16* src = p.q + qi_.row * p.q_stride + head * HD;
#pragma unroll
        for (int ks = 0; ks < G::NKS; ++ks) qf[ks] = *(const typename T::vec8*)(src + 16 * ks + 8 * h);
    };


    // rel-pos tables: the same for every item of this launch
    for (int e = tid; e < 64 * (HD / 4); e += NTHR) {
        const int row = e / (HD / 4), c4 = e % (HD / 4);
        const int tr = row & 31;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tr < 2 * WS - 1) v = *(const f32x4*)((row < 32 ? p.rel_h : p.rel_w) + (size_t)tr * HD + c4 * 4);
        typename T::vec4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
        *(typename T::vec4*)(sTab + row * G::KS + c4 * 8) = o;
    }

    const int Gd = gridDim.x;
    int item = xcd_remap(blockIdx.x, Gd);
    if (item >= nitems) return;
    typename T::vec8 qf[G::NKS], qn[G::NKS];
    prefetch_kv(item);
    load_q(qf, item);
    commit_kv();
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) asm volatile("" : "+v"(qf[ks]));      // landed before the loop, as at its back edge (below)

    v_pad_ones<T, HD>(sV, L::NKEY, tid, NTHR);            // the staging never touches the pad columns again
    __syncthreads();

#if WM_DEV_TIMELINE
    // dev: stamps of workgroup 0 (items 1..3 of its walk), 16 per item: 0 top, 1 prefetch issued, 2 rel-pos U / V ready, 3..6 key steps,
    // 7 stored, 8 barrier, 9 K / V committed, 10 barrier
    unsigned long long* tls = (unsigned long long*)(smem + L::TOTAL) + wave * 64;
    const bool tl_on = p.tl && blockIdx.x == 0;
    int tl_it = 0;
    auto stamp = [&](int k) {
        if (tl_on && tl_it >= 1 && tl_it < 4) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            if (lane == 0) tls[(tl_it - 1) * 16 + k] = t;
        }
    };
#define WM_WIN_STAMP(k) stamp(k)
#else
#define WM_WIN_STAMP(k)
#endif
    while (true) {
        const int next = item + Gd;
        const bool has_next = next < nitems;
        WM_WIN_STAMP(0);
        if (has_next) prefetch_kv(next);                  // in flight during this item's compute
        WM_WIN_STAMP(1);
        // T[c][i]: i<32 -> q.rel_h[i], i>=32 -> q.rel_w[i-32], pre-divided by the softmax scale
        float U[WS], V[WS];
        {
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            qk_tile<T, HD, 2>(acc, qf, sTab, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int il = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                    sT[c * 65 + il] = acc[t][r] * inv_scale;
                }
            // table rows 27..31 are zero, so out-of-window slots (qh, qw up to 15) read zeros
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                U[k] = sT[c * 65 + (qh - k + WS - 1)];
                V[k] = sT[c * 65 + 32 + (qw - k + WS - 1)];
            }
        }
        WM_WIN_STAMP(2);
        SoftmaxState<G::NDT> st;
        st.init();
        // Key slots are laid out 14 rows (kh) x 16 columns (kw; 14 and 15 are zero rows, masked through the bias): a 32-key MFMA
        // tile is 2 kh rows, so for accumulator register r of lane half h the key is kh = 2 (tile) + (r >> 3),
        // kw = (r & 3) + 8 ((r >> 2) & 1) + 4 h: kh is a compile-time constant and kw depends on the lane only through h.
        // Each lane therefore pre-selects its 8 kw values once (Vsel, -1e30 for the two pad columns) and a score's rel-pos
        // bias is ONE add of two registers, U[kh] + Vsel[idx]; the 224 slots are 3 steps of 64 keys + 1 of 32.
        float Vsel[8];
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) {
            const int kw0 = (i8 & 3) + 8 * (i8 >> 2);                              // half 0; half 1: + 4
            Vsel[i8] = h ? (kw0 + 4 < WS ? V[kw0 + 4 < WS ? kw0 + 4 : 0] : -1e30f) : V[kw0 < WS ? kw0 : 0];
        }
        // Key loop in 7 half-steps of 32 keys (two kh rows), software-pipelined inside the wave: QK^T of half-step i + 1 is ISSUED
        // before the exponentials of half-step i, so the matrix pipe works under this wave's own softmax (the two waves of a SIMD
        // overlap only by chance: timeline, 7k cycles of key loop per wave, 14k of a 17k-cycle item on a two-wave SIMD).  Two score
        // tiles of 16 registers alternate -- the same 32 registers the 64-key step held.
        // Scores are log2-domain and relative to st.m (attn16.h "Scores"): q carries c1, and -m rides in the kw bias registers (Vsel),
        // i.e. in the accumulators' initial value, so a probability is exp2 of the accumulator itself (no FMA per score).
        {
            f32x16 sp[2][1];
            auto s_init = [&](f32x16& d, int i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = U[2 * i + (r >> 3)] + Vsel[(r & 3) + 4 * ((r >> 2) & 1)];
            };
            s_init(sp[0][0], 0);
            qk_tile<T, HD, 1>(sp[0], qf, sK, lane);
            const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
            const int v_lane_off = (4 * (g >> 1) + lq) * G::VS + (16 * (g & 1) + 4 * lp) * 2;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                f32x16& cur = sp[i & 1][0];
                float mx0 = -1e30f, mx1 = -1e30f;
#pragma unroll
                for (int r = 0; r < 8; ++r) { mx0 = fmaxf(mx0, cur[r]); mx1 = fmaxf(mx1, cur[8 + r]); }
                float mx = fmaxf(mx0, mx1);
                {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                if (i == 0 || !__all(mx <= RESCALE_THR)) {          // the reference point moves: first half-step, or a maximum grew past the threshold
                    const float d = i == 0 ? mx : fmaxf(mx, 0.f);
                    if (i > 0) {
                        const float alpha = __builtin_amdgcn_exp2f(-d);
                        st.l *= alpha;
#pragma unroll
                        for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) st.o[dt][r] *= alpha;
                    }
                    st.m += d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[r] -= d;
#pragma unroll
                    for (int i8 = 0; i8 < 8; ++i8) Vsel[i8] -= d;   // the following half-steps start from the new reference point
                }
                // From here the order is written out and fenced (sched_barrier): left alone, hipcc clusters the 16 exponentials and
                // puts all 11 MFMAs behind them.  QK^T(i + 1): one MFMA, then three or four scores' exponentials, five times; then
                // P V(i): one MFMA per ~4 vector instructions (the converts of the second P fragment, the next tile's bias sums).
                float ls = 0.f;
                const bool more = i + 1 < 7;
                f32x16& nxt = sp[(i + 1) & 1][0];
                const char* kn = sK + (i + 1) * 32 * G::KS + (lane & 31) * G::KS + 16 * h;
                if (more) s_init(nxt, i + 1);
                typename T::vec8 kf[G::NKS];                                // K fragments: two requested ahead of their MFMA
                if (more) { kf[0] = lds_read_v8<T>(kn); kf[1] = lds_read_v8<T>(kn + 32); }
                __builtin_amdgcn_sched_barrier(0);
                constexpr int EPG = (16 + G::NKS - 1) / G::NKS;             // exponentials per MFMA gap
#pragma unroll
                for (int ks = 0; ks < G::NKS; ++ks) {
                    if (more) {
                        nxt = T::mfma32(kf[ks], qf[ks], nxt);
                        if (ks + 2 < G::NKS) kf[ks + 2] = lds_read_v8<T>(kn + 32 * (ks + 2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = ks * EPG; r < min(16, (ks + 1) * EPG); ++r) {
                        const float pv = __builtin_amdgcn_exp2f(cur[r]);
                        cur[r] = pv;
                        if constexpr (!G::LSUM_IN_O) ls += pv;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                st.l += ls;
                typename T::vec8 pb0, pb1;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pb0[jj] = T::from_f32_bounded(cur[jj]);
                const char* vp = sV + (i * 32) * G::VS + v_lane_off;
                typename T::vec8 va[G::NDT], vb[G::NDT];                    // V^T fragments of the two 16-key halves
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt) va[dt] = lds_read_vT<T>(vp + dt * 64, 8 * G::VS);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt) {
                    st.o[dt] = T::mfma32(va[dt], pb0, st.o[dt]);
                    vb[dt] = lds_read_vT<T>(vp + 16 * G::VS + dt * 64, 8 * G::VS);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jj = dt * 3; jj < min(8, dt * 3 + 3); ++jj) pb1[jj] = T::from_f32_bounded(cur[8 + jj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (G::NDT * 3 < 8) {
#pragma unroll
                    for (int jj = G::NDT * 3; jj < 8; ++jj) pb1[jj] = T::from_f32_bounded(cur[8 + jj]);
                }
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt) st.o[dt] = T::mfma32(vb[dt], pb1, st.o[dt]);
                if (i == 1 || i == 3 || i == 5) WM_WIN_STAMP(3 + i / 2);
            }
        }
        {
            int b, win, head;
            decode(item, b, win, head);
            WM_WIN_STAMP(6);
            // the next item's Q fragments: requested here, where the score / P / bias registers are dead (beside the K / V staging
            // registers they cost 4 spills, and each spill reload's vmcnt(0) serialised the prefetch: timeline), landed by the commit
            if (has_next) load_q(qn, next);
            if (p.out8) {
                const QInfo qo = q_info(item);
                store_out<T, HD>(st, p.out + qo.row * p.out_stride + head * HD, lane, qo.valid, p.out8 + qo.row * p.out_stride + head * HD);
            } else {
                const int wy = win / NWIN, wx = win % NWIN;
                store_out_rows<T, HD>(st, (char*)sT, lane, [&](int r) -> u16* {
                    const int slot = wave * 32 + r;
                    const int sh = slot / WS, sw = slot - sh * WS;
                    const int y = wy * WS + sh, x = wx * WS + sw;
                    const bool ok = slot < NTOK && y < GRID && x < GRID;
                    return ok ? p.out + ((size_t)b * GRID * GRID + (size_t)(y * GRID + x)) * p.out_stride + head * HD : nullptr;
                });
            }
        }
        WM_WIN_STAMP(7);
        if (!has_next) break;
        __syncthreads();                                  // every wave is done with this item's K / V
        WM_WIN_STAMP(8);
        commit_kv();
        WM_WIN_STAMP(9);
        // the Q fragments must have LANDED here: left to hipcc, their vmcnt wait sits at the first MFMA of the next item, behind that
        // item's K / V prefetch in the in-order counter -- the whole prefetch latency exposed at every item start (timeline: 3-4k cycles)
#pragma unroll
        for (int ks = 0; ks < G::NKS; ++ks) { qf[ks] = qn[ks]; asm volatile("" : "+v"(qf[ks])); }
    
        item = next;
        __syncthreads();
        WM_WIN_STAMP(10);
#if WM_DEV_TIMELINE
        ++tl_it;
#endif
    }
#if WM_DEV_TIMELINE
    __syncthreads();
    if (tl_on && lane == 0)
        for (int i = 0; i < 64; ++i) p.tl[wave * 64 + i] = tls[i];
#endif
}

}  // namespace wm
