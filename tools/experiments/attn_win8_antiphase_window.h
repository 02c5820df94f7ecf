// Window attention with the two waves of a SIMD held in ANTI-PHASE (round 4), gfx950.
//
// attn_window_kernel (attn16.h) keeps a whole window's K / V (224 slots) in LDS, stages the next item's through registers and lets
// its 7 waves run QK^T -> softmax -> P V each on its own: the two waves of a SIMD overlap their matrix and vector halves only by
// chance (timeline, round 3: ~6k cycles of key loop per wave, ~12k of an item's 17k on a two-wave SIMD), and ~3.7k cycles per wave and
// item go into issuing the register prefetch and committing it behind two barriers.  This kernel gives the window items the structure
// of attn_global8_kernel (attn_glob8.h):
//   * 8 waves, waves w and w + 4 share a SIMD and are held one phase apart by one workgroup barrier per phase, so one is in a matrix
//     phase (P V of the last key tile, QK^T of the next) while the other is in a vector phase (softmax);
//   * a window's 14 x 16 key slots are four key TILES of 4 kh rows (64 slots; the last tile holds 2 rows = 32 slots), tile t lives in
//     ring slot t, and the NEXT item's tile t is brought in by LDS-DMA as soon as both wave groups are past P V of this item's tile t:
//     no staging registers, no commit pass;
//   * scores are log2-domain (attn16.h "Scores"): the accumulators start from the kw rel-pos term minus the reference point (16
//     registers per lane, the same for both kh rows of an MFMA tile), the kh rel-pos term rides one extra 16-deep k-step of QK^T
//     (B = (hi, lo) pairs of U[kh] for the tile's 4 rows, A = 1.0 at the pair of the key's own row), so a matrix phase holds no vector
//     work and a probability is exp2 of the accumulator itself.
//
// Per item and wave the phases are  M0 V0 M1 V1 M2 V2 M3 V3 M4 E:
//   Mt = P V(t - 1), QK^T(t)   Vt = softmax(t)   E = store the item's output, then the next item's rel-pos prologue (table product,
//   gather, bias fragments).  Wave group 1 runs one phase behind group 0.  Interval i of an item = group 0's phase i:
//     i = 0: all waves issue their DMA pieces of THIS item's tile 3 (its slot was read last in the previous item's interval 9)
//     i = 4 / 6 / 8: the next item's tile 0 / 1 / 2 (read last in this item's interval 3 / 5 / 7)
//   and wait for them (vmcnt(0)) in front of a later barrier: tile 3 at the end of interval 1 (group 0) / of the previous E (group 1), the
//   next item's tiles at the end of E.
// A padded token's K / V row is the 16-bit qkv bias row (image_encoder.py:190-194, 281), the two pad columns of the slot layout take it
// too (their scores carry -1e30 through the kw term, P = 0 exactly): per piece one DMA instruction for the lanes on token rows and one
// for the lanes on bias rows, lane sets by ballot.
// Arithmetic per query as attn_window_kernel (same tiles of 32 keys, same reference-point rule at 64-key granularity): results agree
// with it to the output rounding, not bit for bit (the rule is applied per 64-key tile here, per 32 keys there).
#pragma once
#include "attn_glob8.h"

namespace wm {

template <int HD> struct Window8Lds {
    using G = AttnGeom<HD>;
    static constexpr int NW = 8, NCW = 7;                                   // waves; waves that own queries (7 x 32 >= 196)
    static constexpr int K_BYTES = 64 * G::KS, V_BYTES = 64 * G::VS, TILE = K_BYTES + V_BYTES;
    static constexpr int TAB_BYTES = 64 * G::KS;                           // rel_h rows 0..26, rel_w rows 32..58
    static constexpr int T_BYTES = NCW * 32 * 65 * 4;                      // per computing wave [query][65] fp32 (rel-pos gather, output staging)
    static constexpr int KV_OFF = 0, TAB_OFF = 4 * TILE, T_OFF = TAB_OFF + TAB_BYTES;
    static constexpr int TOTAL = T_OFF + T_BYTES;
    static_assert(K_BYTES % 1024 == 0 && V_BYTES % 1024 == 0, "whole DMA pieces per image");
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <class T, int HD>
__global__ __launch_bounds__(512, 2) void attn_window8_kernel(AttnArgs p, int nitems) {
    using G = AttnGeom<HD>;
    using L = Window8Lds<HD>;
    constexpr int WS = 14, GRID = 64, NWIN = 5, NTOK = WS * WS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                              // SIMD partners: waves w and w + 4
    const bool comp = wave < L::NCW;                        // wave 7 owns no queries: it only moves K / V and keeps the barriers
    // Lane-derived values are re-derived inside every phase from an opaque copy of the lane id: hoisted out of the item loop (hipcc's
    // LICM does that with every address and comparison below) they are ~60 registers that live through all ten phases and spill
    // ("what to look for in the ISA" (2)).
    auto lane_now = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    const int D = p.heads * HD;
    const float inv_scale = 1.0f / p.scale;

    char* sKV = smem + L::KV_OFF;
    char* sTab = smem + L::TAB_OFF;
    float* sT = (float*)(smem + L::T_OFF) + (comp ? wave : 0) * (32 * 65);

    auto decode = [&](int item, int& b, int& win, int& head) {
        head = item % p.heads;
        win = (item / p.heads) % (NWIN * NWIN);
        b = item / (p.heads * NWIN * NWIN);
    };

    // ---- K / V by LDS-DMA.  A tile image is K (64 rows x KS) then V (64 rows x VS); piece q (1 KiB) belongs to wave q % 8.  A lane's 16 B
    // of a piece are (slot row, chunk); slot row = 16 kh_l + kw.  Tile-invariant per lane: the source offset of (kh_l, kw, chunk) inside a
    // window and the geometry word; per tile and item: the window's base address and which lanes sit on token rows / bias rows.
    constexpr int NPK = L::K_BYTES / 1024, NPV = L::V_BYTES / 1024, NPIECE = NPK + NPV;
    constexpr int PER = (NPIECE + 7) / 8;
    static_assert(PER <= 3, "pieces per wave");
    // per lane and piece: `voff` = source offset of (kh_l, kw, chunk) inside a window; `geo` = live | kw << 1 | kh_l << 5 | chunk << 8
    struct Piece { bool isv; unsigned voff, geo; int lds_off; };
    auto piece_setup = [&](int i) {
        Piece d;
        const int q = wave + 8 * i;
        d.isv = q >= NPK;
        const int ql = d.isv ? q - NPK : q;
        const int stride = d.isv ? G::VS : G::KS;
        const int B = ql * 1024 + lane * 16;
        const int row = B / stride, ch = (B % stride) / 16;
        const bool live = q < NPIECE && ch < G::CH;
        d.geo = (live ? 1u : 0u) | ((unsigned)(row & 15) << 1) | ((unsigned)(row >> 4) << 5) | ((unsigned)ch << 8);
        d.voff = live ? (unsigned)((row >> 4) * GRID + (row & 15)) * (unsigned)(d.isv ? p.v_stride : p.k_stride) * 2u + ch * 16u : 0u;
        d.lds_off = (d.isv ? L::K_BYTES : 0) + ql * 1024;
        asm volatile("" : "+v"(d.geo), "+v"(d.voff));      // two registers per piece, not the five values they encode
        return d;
    };
    const Piece pc0 = piece_setup(0), pc1 = piece_setup(1), pc2 = piece_setup(2);
    auto issue1 = [&](const Piece& d, int t, int wy, int wx, const char* kwin, const char* vwin, const char* kbias, const char* vbias, bool pads) {
        unsigned geo = d.geo, voff = d.voff;
        asm volatile("" : "+v"(geo), "+v"(voff));           // decoded here, at every use (see lane_now)
        const int kw = (int)((geo >> 1) & 15u), kh = 4 * t + (int)((geo >> 5) & 7u);       // the window's (kh, kw) of this lane's slot
        const bool in_win = (geo & 1u) && kh < WS;          // tile 3 holds rows 12, 13 only
        const bool tok = in_win && kw < WS && wy * WS + kh < GRID && wx * WS + kw < GRID;
        // bias rows: tokens outside the image; the two pad columns only at start-up (`pads`): their P is exactly 0 whatever finite
        // K / V they hold, and no later DMA touches them
        const unsigned long long m_tok = __ballot(tok), m_bias = __ballot(in_win && !tok && (pads || kw < WS));
        const unsigned dst = (unsigned)(size_t)(lds_cptr_t)(sKV + t * L::TILE + d.lds_off);
        const size_t tile_off = (size_t)(4 * t * GRID) * (size_t)(d.isv ? p.v_stride : p.k_stride) * 2u;
        if (m_tok) dma16_to_lds((d.isv ? vwin : kwin) + tile_off + voff, dst, m_tok);
        if (m_bias) dma16_to_lds((d.isv ? vbias : kbias) + ((geo >> 8) << 4), dst, m_bias);
    };
    auto issue = [&](int item, int t, bool pads = false) {
        int b, win, head;
        decode(item, b, win, head);
        const int wy = win / NWIN, wx = win % NWIN;
        const size_t tok0 = (size_t)b * GRID * GRID + (size_t)(wy * WS * GRID + wx * WS);
        const char* kwin = (const char*)(p.k + tok0 * p.k_stride + head * HD);
        const char* vwin = (const char*)(p.v + tok0 * p.v_stride + head * HD);
        const char* kbias = (const char*)(p.qkv_bias16 + D + head * HD);
        const char* vbias = (const char*)(p.qkv_bias16 + 2 * D + head * HD);
        issue1(pc0, t, wy, wx, kwin, vwin, kbias, vbias, pads);
        if constexpr (PER > 1) issue1(pc1, t, wy, wx, kwin, vwin, kbias, vbias, pads);
        if constexpr (PER > 2) issue1(pc2, t, wy, wx, kwin, vwin, kbias, vbias, pads);
    };

    // ---- this wave's 32 query slots of an item
    struct QInfo { bool valid; size_t row; };
    auto q_info = [&](int item, int ln) {
        const int qi = wave * 32 + (ln & 31);               // slot in the window (0..223; >= 196: none)
        const int qh = qi / WS, qw = qi - qh * WS;
        int b, win, head;
        decode(item, b, win, head);
        const int y = (win / NWIN) * WS + qh, x = (win % NWIN) * WS + qw;
        const bool valid = (qi < NTOK) && (y < GRID) && (x < GRID);
        const size_t tok = valid ? (size_t)(y * GRID + x) : 0;
        return QInfo{valid, (size_t)b * GRID * GRID + tok};
    };
    auto load_q = [&](typename T::vec8 (&q)[G::NKS], int item) {
        const int ln = lane_now();
        int b, win, head;
        decode(item, b, win, head);
        const QInfo qi_ = q_info(item, ln);
        const u16* src = p.q + qi_.row * p.q_stride + head * HD;
#pragma unroll
        for (int ks = 0; ks < G::NKS; ++ks) q[ks] = *(const typename T::vec8*)(src + 16 * ks + 8 * (ln >> 5));
    };

    // rel-pos tables: the same for every item of this launch
    for (int e = tid; e < 64 * (HD / 4); e += 512) {
        const int row = e / (HD / 4), c4 = e % (HD / 4);
        const int tr = row & 31;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tr < 2 * WS - 1) v = *(const f32x4*)((row < 32 ? p.rel_h : p.rel_w) + (size_t)tr * HD + c4 * 4);
        typename T::vec4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
        *(typename T::vec4*)(sTab + row * G::KS + c4 * 8) = o;
    }
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) v_pad_ones<T, HD>(sKV + sl * L::TILE + L::K_BYTES, 64, tid, 512);

    const int Gd = gridDim.x;
    int item = xcd_remap(blockIdx.x, Gd);
    if (item >= nitems) return;

    // per-item state of a computing wave
    typename T::vec8 qf[G::NKS];                            // (the next item's Q is loaded straight into qf in phase 7: QK^T(3), phase 6, was its last use)
    // The item's bias state lives in this wave's staging area (free between the prologue and the output store), 16 B per lane and chunk:
    //   chunks 0..3: the accumulators' initial value `vinit` (kw rel-pos term, pad columns -1e30, minus the reference point): 16 floats
    //   chunks 4..7: the bias k-step's B fragments of key tiles 0..3: (hi, lo) of U[4 T + j], j = 0..3
    // As registers (32 + the next item's Q) they were live through all ten phases and the kernel spilled ~100 registers.
    char* sBias = (char*)sT;
    auto bias_chunk = [&](int ln, int ch) { return sBias + ch * 1024 + ln * 16; };
    SoftmaxState<G::NDT> st;
    f32x16 s[2];
    typename T::vec8 pb[4];

    // rel-pos prologue of an item (needs qf): T[c][i] = q_c . table[i] (i < 32: rel_h rows, i >= 32: rel_w rows), each lane then takes its
    // query's U[kh] = T[qh - kh + 13] and V[kw] = T[32 + qw - kw + 13]
    auto prologue = [&]() {
        const int ln = lane_now();
        const int c = ln & 31, h = ln >> 5;
        const int qi = wave * 32 + c, qh = qi / WS, qw = qi - qh * WS;
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        qk_tile<T, HD, 2>(acc, qf, sTab, ln);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                sT[c * 65 + il] = acc[t][r] * inv_scale;
            }
        float U[16], V[WS];                                 // table rows 27..31 are zero, so out-of-window slots (qh, qw up to 15) read zeros
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            U[k] = sT[c * 65 + (qh - k + WS - 1)];
            V[k] = sT[c * 65 + 32 + (qw - k + WS - 1)];
        }
        U[14] = U[15] = 0.f;
        f32x16 vinit;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kw0 = (r & 3) + 8 * ((r >> 2) & 1);   // accumulator register r of lane half 0; half 1: + 4
            vinit[r] = h ? (kw0 + 4 < WS ? V[kw0 + 4 < WS ? kw0 + 4 : 0] : -1e30f) : V[kw0];
        }
        // (the LDS queue is in order: these writes follow the gathers above)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) *(f32x4*)bias_chunk(ln, ch) = f32x4{vinit[4 * ch], vinit[4 * ch + 1], vinit[4 * ch + 2], vinit[4 * ch + 3]};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            typename T::vec8 bx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typename T::elem hi, lo;
                hi_lo<T>(U[4 * tt + j], hi, lo);
                bx[2 * j] = hi; bx[2 * j + 1] = lo;
            }
            *(typename T::vec8*)bias_chunk(ln, 4 + tt) = bx;
        }
        st.init();
    };

    // ---- phases.  TT = key tile, NT = its 32-key MFMA tiles (2; tile 3: 1)
    auto m_phase = [&](auto tt_c, auto pv_c, auto qk_c) {
        const int ln = lane_now();
        const int h = ln >> 5, r31 = ln & 31;
        const int g4 = ln >> 4, lq = (ln & 15) >> 2, lp = ln & 3;
        const int v_lane_off = (4 * (g4 >> 1) + lq) * G::VS + (16 * (g4 & 1) + 4 * lp) * 2;
        constexpr int TT = decltype(tt_c)::value;
        constexpr bool pv = decltype(pv_c)::value, qk = decltype(qk_c)::value;
        constexpr int NTP = TT - 1 == 3 ? 1 : 2, NTQ = TT == 3 ? 1 : 2;          // MFMA tiles of the P V tile (TT - 1) and of the QK^T tile
        constexpr int AHEAD = 2;
        constexpr int NF_PV = pv ? 2 * NTP * G::NDT : 0, NF_QK = qk ? G::NKS * NTQ : 0;
        if constexpr (pv) {
            const unsigned vbase = lds_base_opaque(sKV + (TT - 1) * L::TILE + L::K_BYTES + v_lane_off);
#pragma unroll
            for (int ks = 0; ks < 2 * NTP; ++ks)
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt) {
                    typename T::vec8 va = lds_read_vT_at<T>(vbase, (16 * ks) * G::VS + dt * 64, 8 * G::VS);
                    st.o[dt] = T::mfma32(va, pb[ks], st.o[dt]);
                }
        }
        if constexpr (qk) {
            f32x16 vinit;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const f32x4 v4 = *(const f32x4*)bias_chunk(ln, ch);
                vinit[4 * ch] = v4[0]; vinit[4 * ch + 1] = v4[1]; vinit[4 * ch + 2] = v4[2]; vinit[4 * ch + 3] = v4[3];
            }
            const typename T::vec8 bx = *(const typename T::vec8*)bias_chunk(ln, 4 + TT);
            const unsigned kbase = lds_base_opaque(sKV + TT * L::TILE + r31 * G::KS + 16 * h);
#pragma unroll
            for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
                for (int t = 0; t < NTQ; ++t) {
                    typename T::vec8 kf = lds_read_v8_at<T>(kbase, 32 * t * G::KS + 32 * ks);
                    s[t] = T::mfma32(kf, qf[ks], ks == 0 ? vinit : s[t]);
                }
#pragma unroll
            for (int t = 0; t < NTQ; ++t)                   // the kh rel-pos term: A = 1.0 at the (hi, lo) pair of the key's own kh row (4 v_cndmask)
                s[t] = T::mfma32(bias_a_frag<T>(2 * t + (r31 >> 4), h == 0), bx, s[t]);
        }
        // fragment reads AHEAD of the MFMA that consumes them (as attn_glob8.h: the SIMD partner is in its vector phase, nobody else
        // covers this wave's LDS latency).  The QK^T part opens with its 5 bias reads (4 x vinit, 1 x bx).
        constexpr int NB = qk ? 5 : 0;
        if constexpr (pv && qk) {
            __builtin_amdgcn_sched_group_barrier(0x100, NB + 2 * AHEAD, 0);
#pragma unroll
            for (int f = 0; f < NF_PV + NF_QK; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + AHEAD < NF_PV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (f + AHEAD < NF_PV + NF_QK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NTQ, 0);
        } else if constexpr (pv) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * AHEAD, 0);
#pragma unroll
            for (int f = 0; f < NF_PV; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + AHEAD < NF_PV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else if constexpr (qk) {
            __builtin_amdgcn_sched_group_barrier(0x100, NB + AHEAD, 0);
#pragma unroll
            for (int f = 0; f < NF_QK; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + AHEAD < NF_QK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NTQ, 0);
        }
        if constexpr (qk) {
#pragma unroll
            for (int t = 0; t < NTQ; ++t) asm volatile("" : "+v"(s[t]));          // the QK^T MFMAs are issued in THIS phase
        }
    };
    auto v_phase = [&](auto tt_c) {
        const int ln = lane_now();
        constexpr int TT = decltype(tt_c)::value;
        constexpr int NT = TT == 3 ? 1 : 2;
        float mx0 = -1e30f, mx1 = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { mx0 = fmaxf(mx0, s[0][r]); mx1 = fmaxf(mx1, s[NT - 1][r]); }
        float mx = fmaxf(mx0, mx1);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (TT == 0 || !__all(mx <= RESCALE_THR)) {         // the reference point moves: first tile, or a maximum grew past the threshold
            const float d = TT == 0 ? mx : fmaxf(mx, 0.f);
            if constexpr (TT > 0) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
                st.l *= alpha;
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st.o[dt][r] *= alpha;
            }
            st.m += d;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] -= d;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {                // the following tiles start from the new reference point
                f32x4 v4 = *(const f32x4*)bias_chunk(ln, ch);
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] -= d;
                *(f32x4*)bias_chunk(ln, ch) = v4;
            }
        }
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[t][r]);
                s[t][r] = pv;
                if constexpr (!G::LSUM_IN_O) ls += pv;
            }
        st.l += ls;
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pb[ks][jj] = T::from_f32_bounded(s[ks >> 1][8 * (ks & 1) + jj]);
        // P must exist HERE (attn_glob8.h: hipcc otherwise sinks the exp / convert chain behind the barrier, into the matrix phase)
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks) asm volatile("" : "+v"(pb[ks]));
    };
    auto store_item = [&](int it) {
        const int ln = lane_now();
        int b, win, head;
        decode(it, b, win, head);
        if (p.out8) {
            const QInfo qo = q_info(it, ln);
            store_out<T, HD>(st, p.out + qo.row * p.out_stride + head * HD, ln, qo.valid, p.out8 + qo.row * p.out_stride + head * HD);
        } else {
            const int wy = win / NWIN, wx = win % NWIN;
            store_out_rows<T, HD>(st, (char*)sT, ln, [&](int r) -> u16* {
                const int slot = wave * 32 + r;
                const int sh = slot / WS, sw = slot - sh * WS;
                const int y = wy * WS + sh, x = wx * WS + sw;
                const bool ok = slot < NTOK && y < GRID && x < GRID;
                return ok ? p.out + ((size_t)b * GRID * GRID + (size_t)(y * GRID + x)) * p.out_stride + head * HD : nullptr;
            });
        }
    };

#if WM_DEV_TIMELINE
    // dev: s_memtime of workgroup 0, third item of its walk: stamp 2 i = phase i's work done (before the barrier), 2 i + 1 = barrier passed
    int tl_it = 0;
    auto stamp = [&](int k) {
        if (p.tl && blockIdx.x == 0 && tl_it == 2) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            if (lane == 0) p.tl[wave * 64 + k] = t;
        }
    };
#define WM_W8_STAMP(k) stamp(k)
#else
#define WM_W8_STAMP(k)
#endif
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto bar_landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // + this wave's DMA pieces (and Q loads) have landed

    // ---- start-up: the first item's four tiles and Q
#pragma unroll
    for (int t = 0; t < 4; ++t) issue(item, t, true);
    if (comp) load_q(qf, item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (comp) {
#pragma unroll
        for (int ks = 0; ks < G::NKS; ++ks) asm volatile("" : "+v"(qf[ks]));
        prologue();
    }

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    // one item, phases 0..9; `first`: the item whose tiles the start-up brought in.  Group 1 runs the same phases one interval later, so
    // its DMA turns and landed-waits sit one phase earlier (interval = phase + grp).
    bool first = true;
    if (grp == 1) bar();                                    // interval 0 of the first item belongs to group 0 alone
    while (true) {
        const int next = item + Gd;
        const bool has_next = next < nitems;
        // E: Q of the next item was requested in phase 7 (two intervals ago); everything this wave has in flight has landed after the wait
        // (the wait comes FIRST: behind the output stores it waited for their acknowledgement too, ~3k cycles in the timeline)
        auto e_phase = [&](bool issue3) {
            if (has_next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (comp) {
#pragma unroll
                    for (int ks = 0; ks < G::NKS; ++ks) asm volatile("" : "+v"(qf[ks]));
                }
                if (issue3) issue(next, 3);                 // group 1: the next item's interval 0
            }
            if (comp) {
                store_item(item);
                if (has_next) prologue();
            }
        };
        if (grp == 0) {
            if (!first) issue(item, 3);                                                     // interval 0
            if (comp) m_phase(I0{}, no, yes); WM_W8_STAMP(0); bar(); WM_W8_STAMP(1);
            if (comp) v_phase(I0{}); WM_W8_STAMP(2); bar_landed(); WM_W8_STAMP(3);                                          // interval 1: tile 3 landed
            if (comp) m_phase(I1{}, yes, yes); WM_W8_STAMP(4); bar(); WM_W8_STAMP(5);
            if (comp) v_phase(I1{}); WM_W8_STAMP(6); bar(); WM_W8_STAMP(7);
            if (has_next) issue(next, 0);                                                   // interval 4
            if (comp) m_phase(I2{}, yes, yes); WM_W8_STAMP(8); bar(); WM_W8_STAMP(9);
            if (comp) v_phase(I2{}); WM_W8_STAMP(10); bar(); WM_W8_STAMP(11);
            if (has_next) issue(next, 1);                                                   // interval 6
            if (comp) m_phase(I3{}, yes, yes); WM_W8_STAMP(12); bar(); WM_W8_STAMP(13);
            if (comp) { v_phase(I3{}); if (has_next) load_q(qf, next); } WM_W8_STAMP(14); bar(); WM_W8_STAMP(15);             // interval 7
            if (has_next) issue(next, 2);                                                   // interval 8
            if (comp) m_phase(I4{}, yes, no); WM_W8_STAMP(16); bar(); WM_W8_STAMP(17);
            e_phase(false);                                                                 // interval 9
            WM_W8_STAMP(18); bar_landed(); WM_W8_STAMP(19);                                                                   // the next item's tiles 0..2 have landed
        } else {
            if (comp) m_phase(I0{}, no, yes); WM_W8_STAMP(0); bar(); WM_W8_STAMP(1);                                        // interval 1
            if (comp) v_phase(I0{}); WM_W8_STAMP(2); bar(); WM_W8_STAMP(3);
            if (comp) m_phase(I1{}, yes, yes); WM_W8_STAMP(4); bar(); WM_W8_STAMP(5);
            if (has_next) issue(next, 0);                                                   // interval 4
            if (comp) v_phase(I1{}); WM_W8_STAMP(6); bar(); WM_W8_STAMP(7);
            if (comp) m_phase(I2{}, yes, yes); WM_W8_STAMP(8); bar(); WM_W8_STAMP(9);
            if (has_next) issue(next, 1);                                                   // interval 6
            if (comp) v_phase(I2{}); WM_W8_STAMP(10); bar(); WM_W8_STAMP(11);
            if (comp) m_phase(I3{}, yes, yes); WM_W8_STAMP(12); bar(); WM_W8_STAMP(13);
            if (has_next) issue(next, 2);                                                   // interval 8
            if (comp) { v_phase(I3{}); if (has_next) load_q(qf, next); } WM_W8_STAMP(14); bar(); WM_W8_STAMP(15);
            if (comp) m_phase(I4{}, yes, no); WM_W8_STAMP(16); bar(); WM_W8_STAMP(17);                                        // interval 9
            e_phase(true);
            WM_W8_STAMP(18); if (has_next) bar_landed(); WM_W8_STAMP(19);                                                     // the next item's tiles 0..3 have landed
        }
        if (!has_next) break;
        item = next;
        first = false;
#if WM_DEV_TIMELINE
        ++tl_it;
#endif
    }
    // (both groups have passed the same number of barriers: group 0 ten per item, group 1 the opening one + ten per item except after its last phase)
}

}  // namespace wm
