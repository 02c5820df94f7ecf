// Window attention, second generation (round 2): the same LDS image and MFMA maps as attn_window_kernel (attn16.h), but
//   * K and V of the NEXT item are brought in by LDS-DMA (global_load_lds, 16 B per lane) instead of through 40 staging
//     registers and a commit pass: K(next) is requested once every wave is past QK^T (barrier A), V(next) once every wave is
//     past P V (barrier B); a wave waits for its own pieces (vmcnt) right before the barrier that also publishes them, so an
//     item costs two barriers and no commit;
//   * exact two-phase softmax: all 224 key slots of a query stay in registers (7 accumulator tiles), so there is no
//     running max / rescale bookkeeping: one max pass, one exp pass, then P V;
//   * key slots are laid out 14 rows x 16 columns (two zero columns masked through the bias), see attn16.h.
// The LDS-DMA writes whole 1 KiB pieces, i.e. also the pad chunks of the K (176 B) and V (192 B) rows; their source is a
// small constant page in global memory (`cpage`, built per launch by attn_win2_cpage_kernel): 16-bit copies of the qkv bias
// (rows of zero-padded tokens, image_encoder.py:190-194,281), a zero chunk and the chunk that holds 1.0 in V's column HD
// (softmax denominators from the matrix pipe, AttnGeom::LSUM_IN_O).
#pragma once
#include "attn16.h"

namespace wm {

// cpage layout (16-bit elements): [0, 3D) bias as q | k | v; [3D, 3D + 64) zeros; [3D + 64, 3D + 72) = {1, 0, 0, 0, 0, 0, 0, 0}; 8 zeros
template <class T>
__global__ __launch_bounds__(256) void attn_win2_cpage_kernel(const float* __restrict__ qkv_bias, u16* __restrict__ cpage, int D3) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < D3 + 80; i += gridDim.x * 256) {
        float v = 0.f;
        if (i < D3) v = qkv_bias[i];
        else if (i == D3 + 64) v = 1.0f;
        const typename T::elem e = T::from_f32(v);
        cpage[i] = __builtin_bit_cast(u16, e);
    }
}

template <int HD> struct Window2Lds {
    using G = AttnGeom<HD>;
    static constexpr int NKEY = 224, NWAVE = 7;
    static constexpr int K_BYTES = NKEY * G::KS, V_BYTES = NKEY * G::VS;          // 39424 + 43008 (hd 80)
    static constexpr int K_PIECES = (K_BYTES + 1023) / 1024, V_PIECES = (V_BYTES + 1023) / 1024;
    static constexpr int K_ALLOC = K_PIECES * 1024, V_ALLOC = V_PIECES * 1024;    // whole pieces
    static constexpr int TAB_BYTES = 64 * G::KS;
    static constexpr int T_BYTES = NWAVE * 32 * 65 * 4;
    static constexpr int K_OFF = 0, V_OFF = K_ALLOC, TAB_OFF = V_OFF + V_ALLOC, T_OFF = TAB_OFF + TAB_BYTES;
    static constexpr int TOTAL = T_OFF + T_BYTES;
    static constexpr int KPW = (K_PIECES + NWAVE - 1) / NWAVE, VPW = (V_PIECES + NWAVE - 1) / NWAVE;   // pieces per wave
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <class T, int HD>
__global__ __launch_bounds__(448, 2) void attn_window2_kernel(AttnArgs p, int nitems, const u16* __restrict__ cpage) {
    using G = AttnGeom<HD>;
    using L = Window2Lds<HD>;
    constexpr int WS = 14, GRID = 64, NWIN = 5, NTOK = WS * WS;
    constexpr int KCH = G::KS / 16, VCH = G::VS / 16;                // 16-byte chunks per K / V row (11 / 12 for hd 80)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int D = p.heads * HD;
    const float c1 = p.scale * 1.44269504088896340736f;
    const float inv_scale = 1.0f / p.scale;

    char* sK = smem + L::K_OFF;
    char* sV = smem + L::V_OFF;
    char* sTab = smem + L::TAB_OFF;
    float* sT = (float*)(smem + L::T_OFF) + wave * (32 * 65);

    auto decode = [&](int item, int& b, int& win, int& head) {
        head = item % p.heads;
        win = (item / p.heads) % (NWIN * NWIN);
        b = item / (p.heads * NWIN * NWIN);
    };

    // ---- LDS-DMA of one item's K or V image.  Piece = 1 KiB of the image; lane -> byte offset 1024 piece + 16 lane -> (row =
    // key slot, chunk).  Chunks >= HD / 8 are the row's pad (K: one chunk; V: the ones chunk and a zero chunk).
    auto dma_image = [&](int item, auto is_v_tag) {
        constexpr bool IS_V = decltype(is_v_tag)::value;
        constexpr int RB = IS_V ? G::VS : G::KS, PIECES = IS_V ? L::V_PIECES : L::K_PIECES, PW = IS_V ? L::VPW : L::KPW;
        constexpr int IMG = IS_V ? L::V_BYTES : L::K_BYTES;
        int b, win, head;
        decode(item, b, win, head);
        const int wy = win / NWIN, wx = win % NWIN;
        const char* tok0 = (const char*)(p.q + ((size_t)b * GRID * GRID) * p.q_stride + head * HD + (IS_V ? 2 * D : D));   // packed qkv: k +D, v +2D
        const char* cb = (const char*)(cpage + (IS_V ? 2 * D : D) + head * HD);
        const char* czero = (const char*)(cpage + 3 * D);
        const char* cone = (const char*)(cpage + 3 * D + 64);
        // lane id from EXEC (v_mbcnt): the per-lane (row, chunk) of every piece is recomputed here per item; derived from a
        // long-lived register hipcc hoists all of it out of the item loop and spills it around the MFMA phases
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = wave + i * L::NWAVE;                   // wave-uniform
            if (piece < PIECES) {
                const int off = piece * 1024 + ln * 16;
                const int row = off / RB, ch = (off - row * RB) >> 4;
                const int kh = row >> 4, kw = row & 15;
                const char* src;
                if (off >= IMG || kw >= WS) src = czero;                                         // beyond the image / pad key slots
                else if (ch >= HD / 8) src = (IS_V && G::LSUM_IN_O && ch == HD / 8) ? cone : czero;   // pad chunks of the row
                else {
                    const int y = wy * WS + kh, x = wx * WS + kw;
                    src = (y < GRID && x < GRID) ? tok0 + ((size_t)(y * GRID + x) * p.q_stride + ch * 8) * 2 : cb + ch * 16;
                }
                __builtin_amdgcn_global_load_lds(src, WM_LDS_PTR((IS_V ? sV : sK) + piece * 1024), 16, 0, 0);
            }
        }
    };
    using IS_K = std::false_type;
    using IS_V = std::true_type;

    // this wave's 32 query slots of an item
    const int qi = wave * 32 + c;
    const int qh = qi / WS, qw = qi - qh * WS;
    auto q_row = [&](int item, bool& valid) {
        int b, win, head;
        decode(item, b, win, head);
        const int y = (win / NWIN) * WS + qh, x = (win % NWIN) * WS + qw;
        valid = (qi < NTOK) && (y < GRID) && (x < GRID);
        return (size_t)b * GRID * GRID + (valid ? (size_t)(y * GRID + x) : 0);
    };
    auto load_q = [&](typename T::vec8 (&qf)[G::NKS], int item) {
        int b, win, head;
        decode(item, b, win, head);
        bool valid;
        const u16* src = p.q + q_row(item, valid) * p.q_stride + head * HD;
#pragma unroll
        for (int ks = 0; ks < G::NKS; ++ks) qf[ks] = *(const typename T::vec8*)(src + 16 * ks + 8 * h);
    };

    // rel-pos tables: the same for every item of this launch
    for (int e = tid; e < 64 * (HD / 4); e += 448) {
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
    load_q(qf, item);
    dma_image(item, IS_K{});
    dma_image(item, IS_V{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    while (true) {
        const int next = item + Gd;
        const bool has_next = next < nitems;
        // ---- rel-pos products T[c][i] and this lane's bias values (as attn_window_kernel)
        float U[WS], Vsel[8];
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
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int k = 0; k < WS; ++k) U[k] = sT[c * 65 + (qh - k + WS - 1)];
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) {
                const int kw = (i8 & 3) + 8 * (i8 >> 2) + 4 * h;                 // this lane's kw for register pattern i8
                Vsel[i8] = kw < WS ? sT[c * 65 + 32 + (qw - (kw < WS ? kw : 0) + WS - 1)] : -1e30f;
            }
        }
        if (has_next) load_q(qn, next);                          // in flight during this item's compute
        // ---- phase 1: all scores S^T = K Q^T + bias, 7 tiles of 32 key slots (2 key rows each)
        f32x16 s[7];
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = U[2 * t + (r >> 3)] + Vsel[(r & 3) + 4 * ((r >> 2) & 1)];
#pragma unroll
        for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                typename T::vec8 kf = lds_read_v8<T>(sK + (32 * t + (lane & 31)) * G::KS + (16 * ks + 8 * h) * 2);
                s[t] = T::mfma32(kf, qf[ks], s[t]);
            }
        // every wave is past QK^T: K may be overwritten; this wave's V pieces (requested an item ago) have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (has_next) dma_image(next, IS_K{});
        // ---- exact softmax over the 224 slots of this lane's query (half of them here, half in lane ^ 32)
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float off = -mx * c1;
        // P^T fragments (16-bit) replace the fp32 scores: 4 registers per 32-key tile and 16-key step
        typename T::vec8 pb[14];
        float lsum = 0.f;                                        // hd 64: no spare O^T row for the denominator
#pragma unroll
        for (int ks = 0; ks < 14; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[ks >> 1][8 * (ks & 1) + j], c1, off));
                pb[ks][j] = T::from_f32(pv);
                if constexpr (!G::LSUM_IN_O) lsum += pv;
            }
        // ---- phase 2: O^T = V^T P^T (row HD of O^T = the denominator, from V's ones column)
        f32x16 o[G::NDT];
#pragma unroll
        for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        {
            const int g = lane >> 4;
            const int lq = (lane & 15) >> 2, lp = lane & 3;
            const int v_lane_off = (4 * (g >> 1) + lq) * G::VS + (16 * (g & 1) + 4 * lp) * 2;
#pragma unroll
            for (int ks = 0; ks < 14; ++ks)
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt) {
                    const char* pv = sV + (16 * ks) * G::VS + dt * 64 + v_lane_off;
                    typename T::vec8 va = lds_read_vT<T>(pv, 8 * G::VS);
                    o[dt] = T::mfma32(va, pb[ks], o[dt]);
                }
        }
        // every wave is past P V: V may be overwritten; this wave's K(next) pieces have landed (and its Q(next) loads: the
        // copy sits here, where everything is drained anyway, not behind the V request below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (has_next) {
#pragma unroll
            for (int ks = 0; ks < G::NKS; ++ks) qf[ks] = qn[ks];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (has_next) dma_image(next, IS_V{});
        // ---- normalise and store
        {
            float l;
            if constexpr (G::LSUM_IN_O) l = __shfl(o[G::NDT - 1][G::LSUM_R], lane & 31, 64);
            else l = lsum + __shfl_xor(lsum, 32, 64);
            const float inv = 1.0f / l;
            int b, win, head;
            decode(item, b, win, head);
            bool valid;
            const size_t row = q_row(item, valid);
            if (valid) {
                u16* orow = p.out + row * p.out_stride + head * HD;
                unsigned char* orow8 = p.out8 ? p.out8 + row * p.out_stride + head * HD : nullptr;
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int d = 32 * dt + 8 * rg + 4 * h;
                        if (d < HD) {
                            if (orow8) {
                                const f32x4 v{o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv};
                                *(unsigned*)(orow8 + d) = pack4_e4m3(v);
                            } else {
                                typename T::vec4 ov;
#pragma unroll
                                for (int j = 0; j < 4; ++j) ov[j] = T::from_f32(o[dt][4 * rg + j] * inv);
                                *(typename T::vec4*)(orow + d) = ov;
                            }
                        }
                    }
            }
        }
        if (!has_next) break;
        item = next;
    }
}

}  // namespace wm
