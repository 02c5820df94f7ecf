// Global attention with the two waves of a SIMD held in ANTI-PHASE (round 3), gfx950.
//
// attn_global_kernel (attn16.h) runs 4-wave workgroups, two per CU: the two waves that share a SIMD belong to different
// workgroups and drift freely, so most of the time both are in the same kind of phase.  Counters (B = 16, hd 80): matrix pipe
// busy 43 % of the SIMD cycles, vector ALU 42 %, both at once 12 %, neither 27 %.  A wave's key tile is a chain -- QK^T (10 MFMAs),
// softmax (~150 vector issue slots), P V (12 MFMAs) -- whose matrix and vector halves take about the same time, so the SIMD's
// partner wave could run its vector half under this wave's matrix half, if the two stayed half a tile apart.  Here they do:
// one 8-wave workgroup per CU, waves 0-3 and 4-7 (SIMD partners) half a tile apart, one workgroup barrier per half tile
// (gemm16_v5.h holds its wave groups apart the same way):
//
//      interval          2j                               2j+1
//      waves 0-3   M: P V(j-1), DMA(j+1), QK^T(j)         V: softmax(j); DMA landed
//      waves 4-7   V: DMA(j+1), softmax(j-1)              M: P V(j-1), QK^T(j); DMA landed
//
// K / V tiles live in a 3-slot LDS ring: tile j is read from interval 2j (waves 0-3, QK^T) to 2j+3 (waves 4-7, P V); tile j+1 goes
// into the slot tile j-2 left at 2j-1.  All eight waves issue their pieces of tile j+1 by LDS-DMA DURING interval 2j and wait for
// them (vmcnt(0)) in front of the barrier that closes interval 2j+1, so a piece has one to two intervals to land.  (First version:
// global -> registers -> ds_write, each thread its own chunks.  Timeline of workgroup 0, s_memtime: the loads' address arithmetic
// and issue cost ~13 % of the kernel, the vmcnt wait + ds_write ~6 %, wherever in the interval they sat.)  The barriers are raw
// s_barrier behind lgkmcnt(0), so a pending DMA never stalls a barrier it need not.  A workgroup covers 256 queries of one
// (image, head).
// Same arithmetic per query as attn_global_kernel (same tile order, same deferred-max rule): bit-identical outputs.
#pragma once
#include <type_traits>
#include "attn16.h"

namespace wm {

// LDS reads at (one opaque base register) + (compile-time offset): with pointers derived from the ring-slot arithmetic hipcc keeps one
// loop-invariant address register PER READ and re-derives each with two vector adds per key tile -- 48 vector instructions in a
// matrix phase that has ~6 issue slots per MFMA, competing with the partner wave's softmax.
typedef __attribute__((address_space(3))) char* lds_cptr_t;
__device__ __forceinline__ unsigned lds_base_opaque(const char* p) {
    unsigned b = (unsigned)(size_t)(lds_cptr_t)p;
    asm volatile("" : "+v"(b));
    return b;
}
// One LDS-DMA piece (16 B per lane, 1 KiB per wave) as inline asm: behind the builtin hipcc cannot tell which later ds_read may touch
// the bytes in flight and puts s_waitcnt vmcnt(0) in front of the next one -- here the QK^T reads of the SAME matrix phase.  The
// waits are this file's (bar_landed).
// `lanes` = the lanes that take part (a piece's pad chunks do not; 0 = this wave has no such piece): EXEC is narrowed inside the asm,
// because a C++ `if` costs two taken branches per piece (~100 cycles per piece in the timeline).
// M0 (the DMA's LDS base) is reserved to the compiler -- a clobber of it is not honoured -- so the asm saves and restores it.
__device__ __forceinline__ void dma16_to_lds(const void* gptr, unsigned lds_addr, unsigned long long lanes) {
    unsigned long long saved;
    unsigned m0_saved;
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_and_saveexec_b64 %0, %4\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1"
                 : "=&s"(saved), "=&s"(m0_saved) : "v"(gptr), "s"(lds_addr), "s"(lanes) : "memory", "scc");
}
template <class T>
__device__ __forceinline__ typename T::vec8 lds_read_v8_at(unsigned base, int off) {
    typedef __attribute__((address_space(3))) const typename T::vec8* lptr;
    return *(lptr)(size_t)(base + off);
}
template <class T>
__device__ __forceinline__ typename T::vec8 lds_read_vT_at(unsigned base, int off, int second_off) {
    typedef __attribute__((address_space(3))) s16x4* lptr;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(size_t)(base + off));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(size_t)(base + off + second_off));
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return __builtin_bit_cast(typename T::vec8, r);
}

template <int HD, bool REL> struct Global8Lds {
    using G = AttnGeom<HD>;
    static constexpr int NW = 8;
    static constexpr int WAVE_F = 32 * 65;                                     // floats per wave (padded staging)
    static constexpr int RELH_BYTES = REL ? NW * WAVE_F * 4 : 0;               // [wave][kh][query] fp32, aliased with [query][65] staging
    static constexpr int K_BYTES = 64 * G::KS, V_BYTES = 64 * G::VS, TILE = K_BYTES + V_BYTES;
    static constexpr int KV_OFF = RELH_BYTES;
    static constexpr int TOTAL = RELH_BYTES + 3 * TILE;
    static_assert(3 * TILE >= 128 * G::KS, "the rel-pos table image is staged in the K/V ring");
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <class T, int HD, bool REL>
__global__ __launch_bounds__(512, 2) void attn_global8_kernel(AttnArgs p) {
    using G = AttnGeom<HD>;
    using L = Global8Lds<HD, REL>;
    constexpr int NT = 2;                                   // 32-key MFMA tiles per key tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                              // SIMD partners are waves w and w + 4 (grouping w, w ^ 1 or w, w ^ 2: 18-20 % slower)
    const int c = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware: the nq / 256 workgroups of one (image, head) read the same K / V, so they get consecutive logical ids =
    // one XCD = one L2 (as a 3-D grid they were dealt round-robin over the 8 XCDs and every L2 fetched every K / V: 1.34 GB of HBM
    // reads per launch for 0.5 GB of qkv, profiles/r3_final_fp16_pmc_traffic.json before this)
    const int nqb = p.nq / 256;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = (lid / nqb) % p.heads, b = lid / (nqb * p.heads);
    const int q0 = (lid % nqb) * 256 + wave * 32;

    const u16* qb = p.q + ((size_t)b * p.nq) * p.q_stride + head * HD;
    const u16* kb = p.k + ((size_t)b * p.nk) * p.k_stride + head * HD;
    const u16* vb = p.v + ((size_t)b * p.nk) * p.v_stride + head * HD;

    typename T::vec8 qf[G::NKS];
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks)
        qf[ks] = *(const typename T::vec8*)(qb + (size_t)(q0 + c) * p.q_stride + 16 * ks + 8 * h);

    char* sKV = smem + L::KV_OFF;
    f32x16 relw[2];
    float* sRelH = (float*)smem + wave * L::WAVE_F;

    // ---- staging by LDS-DMA (tile j lives in ring slot (j + 2) % 3: slot 2 is free while the rel-pos tables occupy slots 0 and 1, so tile 0
    // is requested before the rel-pos arithmetic and lands under it): the K image (64 rows x KS) is KS / 16 pieces of 1 KiB, the V image VS / 16; piece q belongs to wave
    // q % 8.  A lane's 16 B of a piece are (row, chunk) = ((1024 q + 16 lane) / stride, ... % stride / 16); lanes on a pad chunk
    // stay masked, so the V image's ones column (v_pad_ones) survives.  Address = wave-uniform base (K or V, + tile) + a per-lane
    // 32-bit offset computed once: no vector instruction per tile, no staging registers, no ds_write.
    const int ntiles = p.nk / 64;
    constexpr int NPK = G::KS / 16, NPV = G::VS / 16, NPIECE = NPK + NPV;
    constexpr int PER = (NPIECE + 7) / 8;
    static_assert(PER <= 5, "pieces per wave");
    struct Piece { bool isv; unsigned long long lanes; unsigned voff; int lds_off; };
    auto piece_setup = [&](int i) {
        Piece d;
        const int q = wave + 8 * i;
        d.isv = q >= NPK;
        const int ql = d.isv ? q - NPK : q;
        const int stride = d.isv ? G::VS : G::KS;
        const int B = ql * 1024 + lane * 16;
        const int row = B / stride, ch = (B % stride) / 16;
        const bool live = q < NPIECE && ch < G::CH;
        d.lanes = __ballot(live);
        d.voff = live ? (unsigned)row * (unsigned)(d.isv ? p.v_stride : p.k_stride) * 2u + ch * 16u : 0u;
        d.lds_off = (d.isv ? L::K_BYTES : 0) + ql * 1024;
        return d;
    };
    const Piece pc0 = piece_setup(0), pc1 = piece_setup(1), pc2 = piece_setup(2), pc3 = piece_setup(3), pc4 = piece_setup(4);
    auto issue1 = [&](const Piece& d, int tile) {
        const char* base = d.isv ? (const char*)vb + (size_t)tile * 128u * (unsigned)p.v_stride
                                 : (const char*)kb + (size_t)tile * 128u * (unsigned)p.k_stride;
        const unsigned dst = (unsigned)(size_t)(lds_cptr_t)(sKV + ((tile + 2) % 3) * L::TILE + d.lds_off);
        dma16_to_lds(base + d.voff, dst, d.lanes);
    };
    auto issue = [&](int tile) {
        issue1(pc0, tile);
        if constexpr (PER > 1) issue1(pc1, tile);
        if constexpr (PER > 2) issue1(pc2, tile);
        if constexpr (PER > 3) issue1(pc3, tile);
        if constexpr (PER > 4) issue1(pc4, tile);
    };
#if WM_DEV_TIMELINE
    // dev: coarse stamps in slots 60..63 of the wave's timeline: kernel entry, rel-pos prologue done, K / V prologue done, key loop done
    auto stamp_at = [&](int slot) {
        if (p.tl && blockIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            if (lane == 0) ((unsigned long long*)(smem + L::TOTAL) + wave * 64)[slot] = t;
        }
    };
    stamp_at(60);
#define WM_G8_COARSE(slot) stamp_at(slot)
#else
#define WM_G8_COARSE(slot)
#endif
    if constexpr (REL) {
        // ---- prologue: rel_w (registers) and rel_h (LDS) for this wave's 32 queries; as attn_global_kernel ----
        const int qh = q0 >> 6, qw0 = q0 & 63;
        const float inv_scale = 1.0f / p.scale;
        char* sTab = sKV;
        float* sT = (float*)smem + wave * L::WAVE_F;          // [query c][65] fp32 staging
        // both tables staged at once (the K / V ring is idle and holds them side by side), one barrier pair
        static_assert(2 * L::TILE >= 256 * G::KS, "both rel-pos table images are staged in ring slots 0-1 (tile 0 is DMA'd into slot 2 before they are read)");
        __syncthreads();
        {   // all of a thread's table chunks are requested before the first is converted (as a rolled loop each load was waited for
            // in turn: ~12k cycles of the prologue in the timeline)
            constexpr int NTC = 256 * (HD / 4) / 512;
            static_assert(256 * (HD / 4) % 512 == 0, "table chunks per thread");
            f32x4 tv[NTC];
#pragma unroll
            for (int i = 0; i < NTC; ++i) {
                const int e = tid + i * 512, row = e / (HD / 4), c4 = e % (HD / 4);
                const float* tab = row < 128 ? p.rel_w : p.rel_h;
                const int tr = min(row & 127, 126);              // row 127 of an image is zero (below)
                tv[i] = *(const f32x4*)(tab + (size_t)tr * HD + c4 * 4);
            }
#pragma unroll
            for (int i = 0; i < NTC; ++i) {
                const int e = tid + i * 512, row = e / (HD / 4), c4 = e % (HD / 4);
                typename T::vec4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (row & 127) < 127 ? T::from_f32(tv[i][j]) : T::from_f32(0.f);
                *(typename T::vec4*)(sTab + row * G::KS + c4 * 8) = o;
            }
        }
        __syncthreads();
        issue(0);
        WM_G8_COARSE(48);
        {   // rel_w: T[c][i] = q_c . table_w[i] for the 127 rows in two passes of 64; lane (c, h) keeps the entries its keys need
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
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int il = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                        sT[c * 65 + il] = acc[t][r];
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                // every lane reads a (valid) entry and keeps it by select: as `if`s these were 32 divergent branches per pass
                // (8.2k of the prologue's 26k cycles in the timeline)
                float tv[2][16];                               // all 32 reads in flight, then the selects
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kw = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                        tv[t][r] = sT[c * 65 + ((qw + 63 - kw) & 63)];
                    }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kw = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int idx = qw + 63 - kw;
                        asm volatile("" : "+v"(tv[t][r]));     // the read stays unconditional (hipcc sinks it under the condition otherwise)
                        relw[t][r] = (idx >> 6) == pass ? tv[t][r] * inv_scale : relw[t][r];
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
        }
        WM_G8_COARSE(49);
        {   // rel_h: the 64 table rows qh + 63 - kh of this wave's query row, straight from the table image
            const char* sTabH = sTab + 128 * G::KS;
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
                    typename T::vec8 kf = lds_read_v8<T>(sTabH + row * G::KS + (16 * ks + 8 * h) * 2);
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
        WM_G8_COARSE(50);
        __syncthreads();
    }

    WM_G8_COARSE(61);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) v_pad_ones<T, HD>(sKV + sl * L::TILE + L::K_BYTES, 64, tid, 512);
    if constexpr (!REL) issue(0);
    if (ntiles > 1) issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    WM_G8_COARSE(62);
    SoftmaxState<G::NDT> st;
    st.init();
    f32x16 s[NT];
    typename T::vec8 pb[2 * NT];                            // P^T fragments of the tile last softmax-ed
    // the bias k-step's B fragment (attn16.h "Scores"): -m and, with REL, the kh rel-pos terms of 8 tiles; rebuilt every 8 tiles
    // (matrix phase, in front of the tile's QK^T) and when m moves (vector phase, for the next tile's QK^T)
    // (REL instances add their per-tile scalar -- the kh rel-pos term minus m -- per score on the vector pipe instead: with 24 MFMAs
    // per tile the matrix phase became the longer one, 1676 vs 1640 us per launch at B = 16; head_dim 128 without rel-pos gained
    // 14 %, 1257 -> 1086 us, and moved from the 4-wave kernel to this one)
    typename T::vec8 bx;
    auto rebuild = [&]() {
        if constexpr (!REL) bx = bias_b_const<T>(-st.m);
    };
    rebuild();

#if WM_DEV_TIMELINE
    // dev: stamps of workgroup 0, per wave, tiles 4..8, 12 per tile: 0 V start, 1 row max known, 2 P built, 3 staging committed, 4 next loads
    // issued, 5 barrier passed, 6 P V issued, 7 QK^T issued, 8 barrier passed
    unsigned long long* tls = (unsigned long long*)(smem + L::TOTAL) + wave * 64;
    const bool tl_on = p.tl && blockIdx.x == 0;
    auto stamp = [&](int k, int j) {
        if (tl_on && j >= 4 && j < 8) {                    // slots 48..59 belong to the coarse prologue stamps
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            if (lane == 0) tls[(j - 4) * 12 + k] = t;
        }
    };
#define WM_G8_STAMP(k, j) stamp(k, j)
#else
#define WM_G8_STAMP(k, j)
#endif
    // M phase of group-local tile j: P V(j - 1), then QK^T(j).  The wave's SIMD partner is in its vector phase, so nobody else
    // covers this wave's LDS latency: all V^T fragments of P V and all K fragments of QK^T are requested up front (hipcc otherwise
    // puts every ds_read directly in front of its MFMA, ~100 cycles of latency per MFMA; measured 2.1x the whole kernel), and the
    // counted lgkmcnt waits hipcc inserts retire them in order under the MFMAs.
    // (pv / qk are compile-time: with run-time flags hipcc joins the three bodies through copies of the O accumulators, 24 v_mov_b64
    // per tile behind s_nop 7 -- seen in the ISA, 48 cycles per MFMA)
    auto m_phase = [&](int j, auto pv_c, auto qk_c) {
        constexpr bool pv = decltype(pv_c)::value, qk = decltype(qk_c)::value;
        // staging turn of waves 0-3: their pieces of tile j + 1; see the header
        if (grp == 0 && j >= 1 && j + 1 < ntiles) issue(j + 1);
        WM_G8_STAMP(3, j - 1);
        const char* sV = sKV + ((j + 1) % 3) * L::TILE + L::K_BYTES;          // slot of tile j - 1
        const char* sK = sKV + ((j + 2) % 3) * L::TILE;
        const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
        const int v_lane_off = (4 * (g >> 1) + lq) * G::VS + (16 * (g & 1) + 4 * lp) * 2;
        const int r31 = lane & 31;
        constexpr int AHEAD = 2;                             // fragments requested ahead of the MFMA that consumes them (2..8 and
                                                             // s_setprio 3 around the phase: all within 1 %)
        constexpr int NF_PV = 2 * NT * G::NDT, NF_QK = G::NKS * NT;                  // fragments read from LDS (the bias k-step's NT MFMAs read none)
        auto pv_body = [&]() {
            const unsigned vbase = lds_base_opaque(sV + v_lane_off);
#pragma unroll
            for (int ks = 0; ks < 2 * NT; ++ks)
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt) {
                    typename T::vec8 va = lds_read_vT_at<T>(vbase, (16 * ks) * G::VS + dt * 64, 8 * G::VS);
                    st.o[dt] = T::mfma32(va, pb[ks], st.o[dt]);
                }
        };
        auto qk_body = [&]() {
            if constexpr (REL) {
#pragma unroll
                for (int t = 0; t < NT; ++t) s[t] = relw[t];
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            }
            const unsigned kbase = lds_base_opaque(sK + r31 * G::KS + 16 * h);
#pragma unroll
            for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    typename T::vec8 kf = lds_read_v8_at<T>(kbase, 32 * t * G::KS + 32 * ks);
                    s[t] = T::mfma32(kf, qf[ks], s[t]);
                }
            if constexpr (!REL) {                            // the bias k-step: -m from the matrix pipe (attn16.h "Scores")
                const typename T::vec8 ax = bias_a_frag<T>(0, h == 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) s[t] = T::mfma32(ax, bx, s[t]);
            }
        };
        // instruction order of a run of fragments: AHEAD fragments' reads first (a V^T fragment is 2 transposed reads, a K fragment
        // 1), then one MFMA per further fragment's reads -- across the P V / QK^T seam too, so the QK^T reads are not a second exposed
        // LDS latency
        if constexpr (pv && qk) {
            pv_body();
            qk_body();
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * AHEAD, 0);
#pragma unroll
            for (int f = 0; f < NF_PV + NF_QK; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + AHEAD < NF_PV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (f + AHEAD < NF_PV + NF_QK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (!REL) __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        } else if constexpr (pv) {
            pv_body();
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * AHEAD, 0);
#pragma unroll
            for (int f = 0; f < NF_PV; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + AHEAD < NF_PV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else if constexpr (qk) {
            qk_body();
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
            for (int f = 0; f < NF_QK; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + AHEAD < NF_QK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (!REL) __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        }
        WM_G8_STAMP(6, j - 1);
        if constexpr (qk) {
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(s[t]));      // the QK^T MFMAs are issued in THIS phase
        }
    };
    // V phase of tile j: softmax(j) -> pb (and the deferred-max rescale of O), then this thread's staging turn
    auto v_phase = [&](int j) {
        if (grp == 1 && j + 2 < ntiles) issue(j + 2);      // staging turn of waves 4-7: the same interval as waves 0-3's, see the header
        float rh = 0.f;
        if constexpr (REL) rh = sRelH[j * 32 + c];
        float mx0 = -1e30f, mx1 = -1e30f;                   // two chains: a dependent v_max3 issues every ~8 cycles, not 4
#pragma unroll
        for (int r = 0; r < 16; ++r) { mx0 = fmaxf(mx0, s[0][r]); mx1 = fmaxf(mx1, s[NT - 1][r]); }
        float mx = fmaxf(mx0, mx1);
        if constexpr (REL) mx = mx + (rh - st.m);           // REL: the scores lack the kh term and the reference point; !REL: they are relative already
        {   // the other half of the keys sits in lane ^ 32: v_permlane32_swap (vector pipe) instead of ds_bpermute (an LDS round trip)
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
#if WM_DEV_TIMELINE
        asm volatile("" : "+v"(mx));
        WM_G8_STAMP(1, j);
#endif
        // the reference point st.m moves at the first tile and when a maximum grew past the threshold (attn16.h "Scores")
        if (j == 0 || !__all(mx <= RESCALE_THR)) {
            const float d = j == 0 ? mx : fmaxf(mx, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
            st.l *= alpha;
#pragma unroll
            for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st.o[dt][r] *= alpha;
            st.m += d;
            if constexpr (!REL) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[t][r] -= d;
                rebuild();
            }
        }
        const float off = REL ? rh - st.m : 0.f;           // REL: added inside the exp2 expression (hipcc then alternates v_add / v_exp; as a
                                                            // separate loop it emitted 32 adds, then 32 exps: +5 % on the kernel)
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = REL ? __builtin_amdgcn_exp2f(s[t][r] + off) : __builtin_amdgcn_exp2f(s[t][r]);
                s[t][r] = pv;
                if constexpr (!G::LSUM_IN_O) ls += pv;
            }
        // (forcing add / exp to alternate for all 32 scores with sched_group_barrier(VALU, 1) / (TRANS, 1) pairs -- hipcc alternates
        // ~14 pairs and then clusters -- measured 1632 vs 1624 us: not kept)
        st.l += ls;
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pb[ks][jj] = T::from_f32_bounded(s[ks >> 1][8 * (ks & 1) + jj]);
        // P must exist HERE: without the ties hipcc sinks the whole exp / convert chain behind the barrier, to the P V MFMAs that
        // consume it, i.e. into the matrix phase (seen in the ISA; the two phases then run back to back on the SIMD)
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks) asm volatile("" : "+v"(pb[ks]));
        WM_G8_STAMP(2, j);
    };

    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto bar_landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // + this wave's DMA pieces have landed

    // two straight-line loops (one per wave group) rather than one loop that picks the phase per interval: every wave passes
    // 2 ntiles + 2 barriers
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    if (grp == 0) {
        m_phase(0, no, yes); bar();
#pragma unroll 1
        for (int j = 0; j + 1 < ntiles; ++j) {
            WM_G8_STAMP(0, j); v_phase(j); bar_landed(); WM_G8_STAMP(5, j);
            m_phase(j + 1, yes, yes); WM_G8_STAMP(7, j); bar(); WM_G8_STAMP(8, j);
        }
        v_phase(ntiles - 1); bar();
        m_phase(ntiles, yes, no); bar();
        bar();
    } else {
        bar();
        m_phase(0, no, yes); bar();
#pragma unroll 1
        for (int j = 0; j + 1 < ntiles; ++j) {
            WM_G8_STAMP(0, j); v_phase(j); bar(); WM_G8_STAMP(5, j);
            m_phase(j + 1, yes, yes); WM_G8_STAMP(7, j); bar_landed(); WM_G8_STAMP(8, j);
        }
        v_phase(ntiles - 1); bar();
        m_phase(ntiles, yes, no); bar();
    }
    WM_G8_COARSE(63);
#if WM_DEV_TIMELINE
    __syncthreads();
    if (tl_on && lane == 0)
        for (int i = 0; i < 64; ++i) p.tl[wave * 64 + i] = tls[i];
#endif

    u16* orow = p.out + ((size_t)b * p.nq + q0 + c) * p.out_stride + head * HD;
    unsigned char* orow8 = p.out8 ? p.out8 + ((size_t)b * p.nq + q0 + c) * p.out_stride + head * HD : nullptr;
    store_out<T, HD>(st, orow, lane, true, orow8);
}

}  // namespace wm
