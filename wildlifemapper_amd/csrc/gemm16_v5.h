// 16-bit MFMA GEMM, 256 x BN x 32 (BN = 320 | 256), 8 waves x (128 x BN/4): the two wave groups of a
// workgroup run half a K-step apart (gfx950).
//
// gemm16_v3.h runs all 8 waves in lockstep: after each barrier both waves of a SIMD read their fragments and
// issue their LDS-DMA pieces (~100 issue cycles each) at the same time, and the matrix pipe idles meanwhile;
// with all memory traffic removed it still reaches only ~64 % MFMA utilisation.  Here the workgroup's barrier is
// used twice per K-step and the upper wave group (waves 4-7, which share SIMDs with waves 0-3) is shifted by one
// barrier interval:
//
//      interval        X_s -> Y_s                          Y_s -> X_s+1
//      waves 0-3       read fragments of step s,           40 MFMAs of step s
//                      issue DMA pieces of step s+2
//      waves 4-7       40 MFMAs of step s-1                read fragments of step s,
//                                                          issue DMA pieces of step s+2
//
// so each SIMD always has one wave in its MFMA phase while the other reads / issues DMA
// (cdna_hip_programming.md: the 8-phase template's `if (wr == 1) s_barrier`; MI355X_MICROARCH.md "Two waves
// per SIMD", item 9).  Ring, swizzle, counted vmcnt and tile order are gemm16_v3.h's; the epilogue (below, at its
// code) goes through LDS so that global accesses are row-contiguous, and brings an fp32 residual in by LDS-DMA:
//   RAW  a wave waits for its own pieces of step s (vmcnt) before X_s; every read of slot s follows X_s.
//   WAR  slot (s+2)%3 = (s-1)%3 is overwritten after X_s: waves 0-3 read it before Y_s-1, waves 4-7 after
//        Y_s-1 and drain those reads (lgkmcnt(0)) before they arrive at X_s.
// Needs K / 32 >= 2.  No implicit-conv A mode (gemm16_v3.h keeps that).
#pragma once
#include "gemm16_v3.h"

#ifndef WM_GEMM_TIMING_BITS
#define WM_GEMM_TIMING_BITS 0
#endif

namespace wm {

// DBG (dev only, WM_GEMM_DBG=1, see launch_gemm16v5_t): waves 0 and 4 of workgroup 0 record s_memtime at six marks of
// K-steps 8..17, every workgroup its wall-clock entry / first barrier / loop end / stores-acknowledged stamps, into
// p.zero_page.  The instrumented instance is a separate kernel; the product instance carries none of it.
// (LNF, rounds 1-2: the fp32-residual epilogue also LayerNormed the finished rows, the row block's workgroups exchanging their
// statistics in-kernel: zero net gain, superseded by the folded LayerNorm below; tools/experiments/gemm16_v5_lnf_fused_layernorm.h.)
// (A persistent instance -- one workgroup per CU walking its tiles and prefetching the next tile's first K-steps during the
// epilogue -- was built in round 2, bit-identical and 7-15 % slower: tools/experiments/gemm16_v5_persist.h)
// Operand layout (round 3).  A DMA piece is 1 KiB of LDS: 16 rows x 64 B of one K-step.  Read from a row-major operand
// it is 16 separate half lines (64 B each, a row apart); the fabric moves 128-byte lines, and the half-line pieces cost the
// loop 8 % (A/B in one process, profiles/r3_dev/gemm_ab_*pack*).  With `w_packed` / `a_packed` the operand is stored in
// LDS-IMAGE ORDER, [rows / 16][K / 32][64 x 16 B]: position l of a piece holds row l >> 2, chunk (l & 3) ^ ((-(l >> 4)) & 3)
// -- the swizzle already applied -- so a piece is 8 whole lines and the per-lane source offset is lane * 16.  Weights are
// packed once at wm_finalize_weights (pack16_lds_image_kernel); activations are written in this order by their producers
// (LayerNorm, the GELU epilogue below with `out_packed`).  Same bytes in the same LDS places: results are bit-identical.
// (Also measured in round 3 and archived, tools/experiments/gemm16_v5_wdirect.h: W fragments loaded straight into registers,
// bypassing the ring: 1.44x slower from row-major W, 1.07x slower from fragment-packed W.)
//
// Folded LayerNorm (round 3).  The blocks' LayerNorms were a kernel of their own that re-read the fp32 residual stream the
// residual GEMM had just written (6 % of a step).  LN(x) W^T + b = rstd (x (gamma (.) W)^T - mean c1) + c2 with c1[n] = sum_k
// (gamma W)[n][k], c2[n] = sum_k beta[k] W[n][k] + b[n], so the normalisation moves into the CONSUMER's epilogue and the
// statistics into the PRODUCER's:
//   FOLDP instance (proj, lin2, the stem's proj_back): the fp32 + residual epilogue, with a 16-lane group owning a row of the
//     pass, also computes the row's (mean, M2) over this tile's BN columns (two-pass, ln_partial16: the arithmetic of
//     layernorm_tiled_kernel) -> st_stats[m][tile], and writes the finished rows as 16-bit in LDS-image order -> out16 (x16);
//   consumer (qkv, lin1; fold_stats != null): the row block's partials (256 rows x fold_ntile x 8 B, contiguous) come in by
//     one LDS-DMA piece per wave in front of the ring fill, 256 threads combine them (ln_combine, Chan) after the K loop, and
//     the 16-bit epilogue applies rstd (acc - mean c1) + c2 before the activation.
// No workgroup waits for another one (the hand-off is the kernel boundary), unlike the LNF instance.  The operand is the RAW
// residual in 16 bits: its rounding error relative to LN's input scale is what rounding the normalised value costs, as long
// as a row's mean is not large against its spread (it is not: per-token means of the stream are O(0.1 sigma), outlier profile
// included); fp16's range holds for |x| < 65504 (the saturation census watches this buffer).
// (FOLDC, the consumer, is an instance of its own: compiled into the plain instance its extra epilogue state cost that kernel
// 114 spilled registers.)
//
// Split stream (round 4).  The FOLDP epilogue moved 10 bytes per element (fp32 residual in, fp32 out, 16-bit copy out) with every
// CU in the same phase: it is bound by the memory system (proj at 0.31 of peak).  The stream is therefore kept as TWO 16-bit
// planes in LDS-image order, hi = T(x) and lo = fp16(x - hi): x = hi + lo carries 22 (fp16 hi) / 19 (bf16 hi) significant bits,
// the hi plane IS the folded LayerNorm's operand, and the SPLIT instance's epilogue reads 4 and writes 4 bytes per element: per
// pass 2 planes x 2 strips of 16 rows x BN columns = 4 BN / 32 one-KiB pieces by LDS-DMA (the same piece count as the fp32
// tile), v = (acc + bias) + (float(hi) + float(lo)), statistics from v as before, hi' = T(v), lo' = fp16(v - hi').  In place:
// a workgroup reads its own tile's planes before it writes them.  Where a residual GEMM is a half-width launch (1-2 tiles per
// call) the stream stays fp32 and ln_stats_x16_kernel rounds it to hi + lo in place, so a tile's bits do not depend on the batch.
template <class T, int BN, int NSLOT = 3, bool DBG = false, bool FOLDP = false, bool FOLDC = false, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void gemm16v5_kernel(Gemm16Args p) {
    static_assert(!(FOLDC && (FOLDP || DBG)), "the folded-LayerNorm consumer is the plain 16-bit-output kernel");
    static_assert(!SPLIT || FOLDP, "the split-stream epilogue is a form of the statistics-producing one");
    using C = G3<BN, 4>;
    constexpr int AHEAD = NSLOT - 1;                       // K-steps of DMA in flight
    // timing experiments of tools/gemm_bench.py (--act 256 / 512 / 1024): compiled in only with -DWM_GEMM_TIMING_BITS=1
    constexpr bool TB = WM_GEMM_TIMING_BITS != 0;
    const bool dbg_nostore = TB && (p.act & 0x100) != 0, dbg_nodma = TB && (p.act & 0x200) != 0, dbg_noissue = TB && (p.act & 0x400) != 0;
    static_assert(C::W_REM == 0 || C::W_REM == 4, "remainder pieces must fall on one wave group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long wt0 = 0, wt1 = 0, wt2 = 0, mt1 = 0, mt2 = 0, we[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (DBG) wt0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;               // wave group = wr: rows 0-127 / 128-255
    const int fr = lane & 15, fq = lane >> 4;
    const int K = p.K, ns = K / C::BK;
    const char* Ab = (const char*)p.A;
    const char* Wb = (const char*)p.W;

    const int tilesM = p.M / C::BM, tilesN = p.N / BN, total_tiles = tilesM * tilesN;
    auto tile_origin = [&](int tile, int& tm0, int& tn0) {
        const int t = xcd_remap(tile, total_tiles);
        const int gm = p.group_m > 0 ? p.group_m : G16_GROUP_M;
        const int per_group = gm * tilesN;
        const int group = t / per_group;
        const int first_m = group * gm;
        const int gsz = min(gm, tilesM - first_m);
        const int in_group = t - group * per_group;
        tm0 = (first_m + in_group % gsz) * C::BM;
        tn0 = (in_group / gsz) * BN;
    };
    int m0, n0;
    tile_origin(blockIdx.x, m0, n0);

    const unsigned lane_off = (unsigned)(lane >> 2) * (unsigned)(K * 2) + (unsigned)((((lane & 3) ^ ((0 - (lane >> 4)) & 3))) << 4);
    const size_t row_bytes = (size_t)K * 2;
    // DMA pieces of this wave for K-step s into ring slot `slot`; EXTRA: waves 0-3 also carry the remainder W piece
    // The wave-uniform part of a DMA source address is pinned to SGPRs (opaque to the optimiser), so that the instruction takes
    // the saddr form (SGPR pair + 32-bit per-lane offset) instead of a 64-bit per-lane address that is kept as a VGPR pair per
    // piece and advanced with a v_lshl_add_u64 per piece and K-step: 8-10 registers and 4-5 vector instructions per step less.
    auto sgpr_ptr = [](const char* ptr) {
        unsigned long long b = (unsigned long long)ptr;
        asm volatile("" : "+s"(b));
        return (const char*)b;
    };
    auto stage_at = [&](int slot, int s, auto extra_tag, int m0, int n0) {     // (m0, n0): the tile the pieces belong to
        constexpr bool EXTRA = decltype(extra_tag)::value;
        if (dbg_noissue && s >= AHEAD) return;
        if (dbg_nodma) s = 0;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) {
            const int seg = wave * C::A_PIECES + i;
            const bool a_packed = p.a_packed != 0;
            const char* base = a_packed ? sgpr_ptr(Ab + ((size_t)((m0 + seg * 16) / 16) * (size_t)ns + (size_t)s) * 1024)
                                        : sgpr_ptr(Ab + (size_t)(m0 + seg * 16) * row_bytes + (size_t)s * 64);
            __builtin_amdgcn_global_load_lds(base + (a_packed ? (unsigned)lane * 16u : lane_off), WM_LDS_PTR(smem + slot * C::STAGE + seg * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::W_LO + (EXTRA ? 1 : 0); ++i) {
            const int seg = i < C::W_LO ? wave * C::W_LO + i : C::WAVES * C::W_LO + wave;
            const char* base = p.w_packed ? sgpr_ptr(Wb + ((size_t)((n0 + seg * 16) / 16) * (size_t)ns + (size_t)s) * 1024)
                                              : sgpr_ptr(Wb + (size_t)(n0 + seg * 16) * row_bytes + (size_t)s * 64);
            __builtin_amdgcn_global_load_lds(base + (p.w_packed ? (unsigned)lane * 16u : lane_off), WM_LDS_PTR(smem + slot * C::STAGE + C::A_BYTES + seg * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](int slot, int s, auto extra_tag) { stage_at(slot, s, extra_tag, m0, n0); };

    const int frag_off = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);
    const int rd_a_k = (wr * 128) * 64 + frag_off;
    const int rd_w_k = C::A_BYTES + (wc * C::WCOLS) * 64 + frag_off;

    f32x4 acc[C::MT][C::NT];
    typename T::vec8 wf[C::NT], af[C::MT];

    auto read_frags = [&](int slot) {
        const char* sS = smem + slot * C::STAGE;
        const int rd_a = rd_a_k, rd_w = rd_w_k;
        af[0] = *(const typename T::vec8*)(sS + rd_a);
#pragma unroll
        for (int i = 0; i < C::NT; ++i) wf[i] = *(const typename T::vec8*)(sS + rd_w + i * 1024);
#pragma unroll
        for (int i = 1; i < C::MT; ++i) af[i] = *(const typename T::vec8*)(sS + rd_a + i * 1024);
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) acc[mi][ni] = T::mfma16(wf[ni], af[mi], acc[mi][ni]);
    };
    auto inc = [](int v) { return v == NSLOT - 1 ? 0 : v + 1; };
    auto dec = [](int v) { return v == 0 ? NSLOT - 1 : v - 1; };
    constexpr int RP_CPR = BN * 4 / 16, RP_PW = RP_CPR / 16;       // residual tile: 16-byte chunks per row; DMA pieces per wave and pass
    auto wait_step = [&](int s, auto extra_tag) {          // this wave's pieces of step s have landed
        constexpr int P = C::P_LO + (decltype(extra_tag)::value ? 1 : 0);
        if (dbg_noissue) { wait_vmcnt<0>(); return; }
        if constexpr (AHEAD == 3) {
            if (s + 2 < ns) { wait_vmcnt<2 * P>(); return; }
        }
        if (s + 1 < ns) wait_vmcnt<P>(); else wait_vmcnt<0>();
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    unsigned tmark = 0;
    auto mark = [&](int s, int k) {
        if constexpr (DBG) {
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            const int idx = (s - 8) * 6 + k;
            tmark = (lane == idx) ? t : tmark;
        }
    };

    // fp32 residual (proj / lin2: out = residual + A W^T + b): the residual tile comes in by LDS-DMA, 32 rows (16 per
    // wave group) per epilogue pass, one pass ahead; pass 0 is requested at the start of the tile and lands beside the ring.
    constexpr int RES_BYTES = 32 * BN * 4;
    constexpr int EPI_BASE = 0;                                                   // staging of the 16-bit / fp32 epilogues
    constexpr int RES_STG = 0;                                                    // staging of the residual epilogue
    constexpr int RES_L1 = 45056;
    constexpr int RES_L0 = NSLOT * C::STAGE;
    constexpr int LDS_TOTAL = NSLOT * C::STAGE + RES_BYTES;
    static_assert(RES_STG + 32 * (BN * 4 + 16) <= RES_L1 && RES_L0 >= NSLOT * C::STAGE && RES_L0 + RES_BYTES <= LDS_TOTAL &&
                  RES_L1 + RES_BYTES <= NSLOT * C::STAGE && RES_L1 % 16 == 0 && RES_L0 % 16 == 0,
                  "epilogue LDS map");
    const int res_mod = p.res_mod > 0 ? p.res_mod : p.M;
    const bool res_wrap = res_mod != p.M;                           // broadcast residual (pos_embed): row m % res_mod
    constexpr int SP_KT = BN / 32;                                 // split stream: pieces per 16-row strip and plane
    static_assert(4 * SP_KT == C::WAVES * RP_PW, "split-stream pieces per pass = the fp32 tile's");
    auto res_dma = [&](int q) {
        char* dst = smem + ((q & 1) ? RES_L1 : RES_L0);
        if constexpr (SPLIT) {
            // piece idx: plane (hi | lo) x strip (rows q*16.. of the upper | lower 128) x kt; a piece is 1 KiB contiguous in memory
#pragma unroll
            for (int i = 0; i < RP_PW; ++i) {
                const int idx = wave * RP_PW + i;
                const int plane = idx / (2 * SP_KT), strip = (idx / SP_KT) & 1, kt = idx % SP_KT;
                const char* src = (const char*)(plane ? p.res_lo : p.res_hi);
                const char* base = sgpr_ptr(src + ((size_t)((m0 + strip * 128 + q * 16) >> 4) * (size_t)(p.N >> 5) + (size_t)((n0 >> 5) + kt)) * 1024);
                __builtin_amdgcn_global_load_lds(base + (unsigned)lane * 16u, WM_LDS_PTR(dst + idx * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
        for (int i = 0; i < RP_PW; ++i) {
            const int piece = wave * RP_PW + i;
            const int c = piece * 64 + lane, r = c / RP_CPR, ch = c - r * RP_CPR;
            int m = m0 + (r >> 4) * 128 + q * 16 + (r & 15);
            if (res_wrap) m %= res_mod;
            __builtin_amdgcn_global_load_lds((const char*)(p.residual + (size_t)m * p.N + n0 + ch * 4), WM_LDS_PTR(dst + piece * 1024), 16, 0, 0);
        }
        }
    };

#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (SPLIT || p.residual) res_dma(0);
    // landing buffer 0 is idle without a residual: raw partials (8 KiB), (mean, rstd) per row (2 KiB), c1 | c2 of the tile's columns
    constexpr int FOLD_RAW = LDS_TOTAL - RES_BYTES, FOLD_MR = FOLD_RAW + 8192, FOLD_C = FOLD_MR + 2048;
    static_assert(FOLD_C + 2 * BN * 4 <= LDS_TOTAL, "fold LDS map");
    if constexpr (FOLDC) {
        // the row block's partial statistics: 256 rows x fold_ntile x (mean, M2), contiguous; one DMA piece per wave
        if (wave * 1024 < 256 * p.fold_ntile * 8)
            __builtin_amdgcn_global_load_lds((const char*)(p.fold_stats + (size_t)m0 * p.fold_ntile * 2) + wave * 1024 + lane * 16,
                                             WM_LDS_PTR(smem + FOLD_RAW + wave * 1024), 16, 0, 0);
        // c1 and c2 (= the bias argument) of this tile's BN columns: 256-byte pieces (4 B per lane: exact, no over-read), into LDS
        // rather than 40 registers per lane -- the epilogue keeps the 160 accumulators live and reads them per column fragment
        constexpr int PC = BN * 4 / 256;
#pragma unroll
        for (int pi = wave; pi < 2 * PC; pi += C::WAVES) {
            const float* src = (pi < PC ? p.fold_c1 + n0 + pi * 64 : p.bias + n0 + (pi - PC) * 64) + lane;
            __builtin_amdgcn_global_load_lds(src, WM_LDS_PTR(smem + FOLD_C + (pi < PC ? pi * 256 : BN * 4 + (pi - PC) * 256)), 4, 0, 0);
        }
    }

    if (wr == 0) {
        using EX = std::integral_constant<bool, (C::W_REM > 0)>;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            if (i < ns) stage(i, i, EX{});
        int slot = 0;
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            mark(s, 0);
            wait_step(s, EX{});
            mark(s, 1);
            barrier();                                      // X_s
            if constexpr (DBG) { if (s == 0) { wt1 = wall_clock64(); mt1 = __builtin_readcyclecounter(); } }
            mark(s, 2);
            read_frags(slot);
            if (s + AHEAD < ns) stage(dec(slot), s + AHEAD, EX{});
            mark(s, 3);
            barrier();                                      // Y_s
            mark(s, 4);
            mfmas();
            mark(s, 5);
            slot = inc(slot);
        }
    } else {
        using EX = std::false_type;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            if (i < ns) stage(i, i, EX{});
        int slot = 0;
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of the slot about to be overwritten are back
            mark(s, 0);
            wait_step(s, EX{});
            mark(s, 1);
            barrier();                                      // X_s
            mark(s, 2);
            if (s > 0) mfmas();                             // step s-1
            mark(s, 3);
            barrier();                                      // Y_s
            mark(s, 4);
            read_frags(slot);
            if (s + AHEAD < ns) stage(dec(slot), s + AHEAD, EX{});
            mark(s, 5);
            slot = inc(slot);
        }
        mfmas();                                            // step ns-1
    }

    if constexpr (DBG) {
        if (blockIdx.x == 0 && (wave == 0 || wave == 4)) ((unsigned*)p.zero_page)[wr * 64 + lane] = tmark;
        wt2 = wall_clock64();
        mt2 = __builtin_readcyclecounter();
    }
    // ---- epilogue: bias / activation in registers, then through LDS so that every global access is row-contiguous.
    // A lane holds C[m = fr][n = 4 fq .. 4 fq + 3] of each 16x16 fragment: stored directly, the 64 lanes of one
    // instruction hit 64 separate 8-byte pieces (16 rows x 4), which the CU's store path takes one at a time
    // (measured 13 us per 256x320 tile, a quarter of the kernel).  Staged through the (now idle) ring in passes of
    // 128 rows (16-bit output) or 64 rows (fp32 output), each thread then moves 16-byte chunks that are consecutive
    // along the row, and the residual is read the same way.
    const int act = p.act & 0xff;
    const int tid_e = tid, fr_e = fr, fq_e = fq;
    f32x4 bias_v[FOLDC ? 1 : C::NT];
    if constexpr (!FOLDC) {
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni)
            bias_v[ni] = p.bias ? *(const f32x4*)(p.bias + n0 + wc * C::WCOLS + ni * 16 + fq_e * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (FOLDC) {
        {
            if (tid < C::BM) {                              // one row per thread: Chan combination of its column-tile partials
                const float2* raw = (const float2*)(smem + FOLD_RAW) + tid * p.fold_ntile;
                float mk[8], qk[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    mk[i] = qk[i] = 0.f;
                    if (i < p.fold_ntile) { const float2 t = raw[i]; mk[i] = t.x; qk[i] = t.y; }
                }
                float rstd;
                const float mean = ln_combine(mk, qk, p.fold_ntile, p.fold_bn, (float)p.K, p.fold_eps, rstd);
                *(float2*)(smem + FOLD_MR + tid * 8) = make_float2(mean, rstd);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): this wave's fragment reads are back
    barrier();                                              // every wave is done with the ring
    if constexpr (DBG) we[0] = wall_clock64();
    // (the epilogue's staging never touches FOLD_RAW / FOLD_MR / FOLD_C)
    // one straight-line instance per activation (a per-fragment runtime branch costs more than the stores)
    auto epilogue = [&](auto act_tag, auto fold_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    constexpr bool FOLD = decltype(fold_tag)::value && FOLDC;
    const int tid = tid_e, fr = fr_e, fq = fq_e;           // shadow the kernel-scope values (see tid_e)
    auto finish = [&](f32x4 v, int ni) {
        if constexpr (!FOLD) v += bias_v[ni];              // FOLD: the caller has applied rstd (acc - mean c1) + c2
        if constexpr (ACT == ACT_GELU) {
            v = gelu_erf_fast4(v);
        } else if constexpr (ACT == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        return v;
    };
    if constexpr (FOLDP) {
        // ---- fp32 + residual, per-row partial LayerNorm statistics, 16-bit copy in LDS-image order (see "Folded LayerNorm") ----
        constexpr int ROWB = BN * 4 + 16, CPT = RP_CPR / 16;
        const int r = tid >> 4, l16 = tid & 15;              // row of the pass, position in the row's 16-lane group
        const int ntile = p.N / BN, nt = n0 / BN;
#pragma unroll
        for (int q = 0; q < C::MT; ++q) {
            if (q + 1 < C::MT) res_dma(q + 1);
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni)
                *(f32x4*)(smem + RES_STG + (wr * 16 + fr) * ROWB + (wc * C::WCOLS + ni * 16 + fq * 4) * 4) = finish(acc[q][ni], ni);
            if (q + 1 < C::MT) wait_vmcnt<RP_PW>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            barrier();
            const char* land = smem + ((q & 1) ? RES_L1 : RES_L0);
            const int m = m0 + (r >> 4) * 128 + q * 16 + (r & 15);
            f32x4 v[CPT];
            float amax = 0.f;
#pragma unroll
            for (int kk = 0; kk < CPT; ++kk) {
                const int ch = l16 + 16 * kk;
                f32x4 res;
                if constexpr (SPLIT) {
                    // the landed planes are in LDS-image order: piece (strip, kt), position = row * 4 + swizzled chunk
                    const int rr = r & 15, kt = ch >> 3, pos = rr * 4 + ((((ch & 7) >> 1)) ^ ((0 - (rr >> 2)) & 3));
                    const int off = (((r >> 4) * SP_KT + kt) << 10) + pos * 16 + (ch & 1) * 8;
                    const typename T::vec4 h4 = *(const typename T::vec4*)(land + off);
                    const f16x4 l4 = *(const f16x4*)(land + 2 * SP_KT * 1024 + off);
#pragma unroll
                    for (int j = 0; j < 4; ++j) res[j] = T::to_f32(h4[j]) + (float)l4[j];
                } else {
                    res = *(const f32x4*)(land + (r * RP_CPR + ch) * 16);
                }
                v[kk] = *(const f32x4*)(smem + RES_STG + r * ROWB + ch * 16) + res;
                if (p.out32) *(f32x4*)(p.out32 + (size_t)m * p.N + n0 + ch * 4) = v[kk];
                if constexpr (std::is_same<T, FP16>::value) amax = fmaxf(fmaxf(fmaxf(fabsf(v[kk][0]), fabsf(v[kk][1])), fmaxf(fabsf(v[kk][2]), fabsf(v[kk][3]))), amax);
            }
            float mean, m2;
            ln_partial16<CPT>(v, 1.0f / BN, mean, m2);
            if (l16 == 0) *(float2*)(p.st_stats + ((size_t)m * ntile + nt) * 2) = make_float2(mean, m2);
            if constexpr (std::is_same<T, FP16>::value) {
                if (amax >= 65504.f && p.overflow) *(volatile int*)p.overflow = 1;     // the fp16 plane clamps: loud on the host side (wm_overflow_count)
            }
            // a wave holds 4 consecutive rows: per K-step of the copy its 16 lanes x 4 rows write 256 contiguous bytes
#pragma unroll
            for (int kk = 0; kk < CPT; ++kk) {
                typename T::vec4 o;
                f16x4 lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = T::from_f32(v[kk][j]);
                    lo[j] = FP16::from_f32(v[kk][j] - T::to_f32(o[j]));
                }
                const int64_t e = lds_image_index(m, n0 + (l16 + 16 * kk) * 4, p.N);
                *(typename T::vec4*)(p.out16 + e) = o;
                if (SPLIT || p.out_lo) *(f16x4*)(p.out_lo + e) = lo;
            }
            if (q + 1 < C::MT) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                barrier();
            }
        }
    } else if (p.out32 == nullptr && p.residual == nullptr) {
        // 16-bit staging: 2 passes of 4 row-fragments; LDS row = BN * 2 + 16 bytes
        constexpr int ROWB = BN * 2 + 16, CPR = BN * 2 / 16, MTP = 4, ROWS = 2 * MTP * 16;
        static_assert(EPI_BASE + ROWS * ROWB <= LDS_TOTAL && (ROWS * CPR) % 512 == 0, "epilogue staging");
        auto epi_sync = [&]() { __syncthreads(); };
#pragma unroll
        for (int q = 0; q < C::MT / MTP; ++q) {
            if constexpr (FOLD) {
                // LayerNorm folded into this GEMM: rstd (acc - mean c1) + c2.  (mean, rstd) of the pass's MTP rows and, per column
                // fragment, c1 / c2 are read from LDS once and reused (28 reads per lane and tile instead of 120)
                float2 mr[MTP];
#pragma unroll
                for (int mm = 0; mm < MTP; ++mm) mr[mm] = *(const float2*)(smem + FOLD_MR + (wr * 128 + (q * MTP + mm) * 16 + fr) * 8);
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {
                    const f32x4 c1 = *(const f32x4*)(smem + FOLD_C + (wc * C::WCOLS + ni * 16 + fq * 4) * 4);
                    const f32x4 c2 = *(const f32x4*)(smem + FOLD_C + BN * 4 + (wc * C::WCOLS + ni * 16 + fq * 4) * 4);
#pragma unroll
                    for (int mm = 0; mm < MTP; ++mm) {
                        f32x4 v = acc[q * MTP + mm][ni];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaf(fmaf(-mr[mm].x, c1[j], v[j]), mr[mm].y, c2[j]);
                        v = finish(v, ni);
                        typename T::vec4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                        *(typename T::vec4*)(smem + EPI_BASE + (wr * MTP * 16 + mm * 16 + fr) * ROWB + (wc * C::WCOLS + ni * 16 + fq * 4) * 2) = o;
                    }
                }
            } else {
#pragma unroll
            for (int mm = 0; mm < MTP; ++mm)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {
                    const f32x4 v = finish(acc[q * MTP + mm][ni], ni);
                    typename T::vec4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                    *(typename T::vec4*)(smem + EPI_BASE + (wr * MTP * 16 + mm * 16 + fr) * ROWB + (wc * C::WCOLS + ni * 16 + fq * 4) * 2) = o;
                }
            }
            epi_sync();
            if constexpr (DBG) { if (q < 2) we[1 + 2 * q] = wall_clock64(); }
            if (p.out_packed) {
                // the output is the next GEMM's A operand: written in LDS-image order.  A wave moves one 16 x 32 tile (1 KiB,
                // contiguous in memory) per iteration: lane l = position l of the piece = row l >> 2, logical chunk
                // (l & 3) ^ ((-(l >> 4)) & 3) of the staged rows.
                constexpr int KT = BN / 32;
                static_assert((ROWS / 16) * KT * 64 == ROWS * CPR, "packed epilogue");
#pragma unroll
                for (int it = 0; it < ROWS * CPR / 512; ++it) {
                    const int tile = it * 8 + wave, rt = tile / KT, kt = tile - rt * KT;
                    const int l = tid & 63, lc = (l & 3) ^ ((0 - (l >> 4)) & 3);
                    const f32x4 v = *(const f32x4*)(smem + EPI_BASE + (rt * 16 + (l >> 2)) * ROWB + (kt * 4 + lc) * 16);
                    const int mt = (m0 + (rt / MTP) * 128 + q * MTP * 16 + (rt % MTP) * 16) >> 4;
                    if (!dbg_nostore) *(f32x4*)((char*)p.out16 + ((size_t)mt * (p.N >> 5) + (n0 >> 5) + kt) * 1024 + l * 16) = v;
                }
            } else {
#pragma unroll
            for (int it = 0; it < ROWS * CPR / 512; ++it) {
                const int c = it * 512 + tid, r = c / CPR, ch = c - r * CPR;
                const int m = m0 + (r / (MTP * 16)) * 128 + q * MTP * 16 + (r % (MTP * 16));
                const f32x4 v = *(const f32x4*)(smem + EPI_BASE + r * ROWB + ch * 16);
                if (!dbg_nostore) *(f32x4*)((char*)p.out16 + ((size_t)m * p.N + n0) * 2 + ch * 16) = v;
            }
            }
            if constexpr (DBG) { if (q < 2) we[2 + 2 * q] = wall_clock64(); }
            if (q + 1 < C::MT / MTP) epi_sync();
        }
    } else if (p.residual != nullptr) {
        // fp32 + residual: 8 passes of one row-fragment (32 rows).  Pass q: request the residual rows of pass q + 1,
        // stage this pass's accumulators, wait for this wave's residual pieces of pass q (everything but the DMA
        // just issued is complete: the C stores of pass q - 1 are older and have had a pass to be acknowledged),
        // barrier, then add and store 16-byte chunks along the rows.
        constexpr int ROWB = BN * 4 + 16, NIT = 32 * RP_CPR / 512;
        static_assert((32 * RP_CPR) % 512 == 0, "epilogue staging");
#pragma unroll
        for (int q = 0; q < C::MT; ++q) {
            if (q + 1 < C::MT) res_dma(q + 1);
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni)
                *(f32x4*)(smem + RES_STG + (wr * 16 + fr) * ROWB + (wc * C::WCOLS + ni * 16 + fq * 4) * 4) = finish(acc[q][ni], ni);
            if (q + 1 < C::MT) wait_vmcnt<RP_PW>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): staging writes done
            barrier();
            const char* land = smem + ((q & 1) ? RES_L1 : RES_L0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 512 + tid, r = c / RP_CPR, ch = c - r * RP_CPR;
                const int m = m0 + (r >> 4) * 128 + q * 16 + (r & 15);
                const f32x4 v = *(const f32x4*)(smem + RES_STG + r * ROWB + ch * 16) + *(const f32x4*)(land + c * 16);
                if (dbg_nostore) { if (v[0] == 12345.678f) p.out16[0] = 1; continue; }
                if (p.out32) *(f32x4*)(p.out32 + (size_t)m * p.N + n0 + ch * 4) = v;
                if (p.out16) {
                    typename T::vec4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                    *(typename T::vec4*)(p.out16 + (size_t)m * p.N + n0 + ch * 4) = o;
                }
            }
            if (q + 1 < C::MT) {
                __builtin_amdgcn_s_waitcnt(0xc07f);         // the LDS reads of this pass are back before anyone overwrites
                barrier();
            }
        }
    } else {
        // fp32 staging: 4 passes of 2 row-fragments; LDS row = BN * 4 + 16 bytes
        constexpr int ROWB = BN * 4 + 16, CPR = BN * 4 / 16, MTP = 2, ROWS = 2 * MTP * 16;
        static_assert(EPI_BASE + ROWS * ROWB <= LDS_TOTAL && (ROWS * CPR) % 512 == 0, "epilogue staging");
        auto epi_sync = [&]() { __syncthreads(); };
#pragma unroll
        for (int q = 0; q < C::MT / MTP; ++q) {
#pragma unroll
            for (int mm = 0; mm < MTP; ++mm)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni)
                    *(f32x4*)(smem + EPI_BASE + (wr * MTP * 16 + mm * 16 + fr) * ROWB + (wc * C::WCOLS + ni * 16 + fq * 4) * 4) = finish(acc[q * MTP + mm][ni], ni);
            epi_sync();
#pragma unroll
            for (int it = 0; it < ROWS * CPR / 512; ++it) {
                const int c = it * 512 + tid, r = c / CPR, ch = c - r * CPR;
                const int m = m0 + (r / (MTP * 16)) * 128 + q * MTP * 16 + (r % (MTP * 16));
                f32x4 v = *(const f32x4*)(smem + EPI_BASE + r * ROWB + ch * 16);
                if (dbg_nostore) { if (v[0] == 12345.678f) p.out16[0] = 1; continue; }
                if (p.out32) *(f32x4*)(p.out32 + (size_t)m * p.N + n0 + ch * 4) = v;
                if (p.out16) {
                    typename T::vec4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                    *(typename T::vec4*)(p.out16 + (size_t)m * p.N + n0 + ch * 4) = o;
                }
            }
            if (q + 1 < C::MT / MTP) epi_sync();
        }
    }
    };
    if constexpr (FOLDP) {                                    // (SPLIT is a form of it)
        epilogue(std::integral_constant<int, ACT_NONE>{}, std::false_type{});
    } else if constexpr (FOLDC) {                            // folded LayerNorm: the 16-bit-output GEMMs qkv (no activation) and lin1 (GELU)
        if (act == ACT_GELU) epilogue(std::integral_constant<int, ACT_GELU>{}, std::true_type{});
        else epilogue(std::integral_constant<int, ACT_NONE>{}, std::true_type{});
    } else {
        if (act == ACT_GELU) epilogue(std::integral_constant<int, ACT_GELU>{}, std::false_type{});
        else if (act == ACT_RELU) epilogue(std::integral_constant<int, ACT_RELU>{}, std::false_type{});
        else epilogue(std::integral_constant<int, ACT_NONE>{}, std::false_type{});
    }
    if constexpr (DBG) {
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // C stores of this wave acknowledged
            const unsigned long long wt3 = wall_clock64();
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            if (lane == 0) {
                unsigned long long* r = (unsigned long long*)((char*)p.zero_page + 512) + (size_t)blockIdx.x * 5;
                r[0] = wt0; r[1] = wt1; r[2] = wt2; r[3] = wt3; r[4] = ((unsigned long long)xcc << 32) | hw;
                if (blockIdx.x < 8) {
                    unsigned* e = (unsigned*)p.zero_page + 64 + 40;          // unused tail of the group-1 marks
                    if (blockIdx.x == 0) {
                        for (int i = 0; i < 5; ++i) e[i] = (unsigned)(we[i] - wt2);
                        e[5] = (unsigned)(mt2 - mt1); e[6] = (unsigned)(wt2 - wt1);      // loop: s_memtime counts, 10 ns units
                    }
                }
            }
        }
    }
}

}  // namespace wm
