// fp8 (OCP e4m3) MFMA GEMM for the transformer blocks' four projections (BASELINE.json configs[4]): the block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales runs at twice the bf16 rate on gfx950 (MI355X_MICROARCH.md
// "Matrix cores"; the non-scaled fp8 MFMAs run at the bf16 rate).
//
//   C[M,N] = act((A[M,K] W[N,K]^T) * wscale[n] + bias[n]) (+ residual[M,N])
//   A: e4m3, unit scale (activations: LayerNorm / attention / GELU outputs, saturated to +-448 when they are produced)
//   W: e4m3 with one fp32 scale per output channel (absmax / 448, computed when the weights are packed)
//   outputs: 16-bit (qkv -> the bf16 attention kernels) | e4m3 (lin1 + GELU -> lin2) | fp32 + fp32 residual (proj, lin2)
//
// Structure = gemm16_v5.h (LDS ring filled by LDS-DMA, XOR swizzle on the source and on the read address, counted vmcnt +
// raw s_barrier, the two wave groups of the workgroup half a K-step apart), with what fp8 changes:
//   * 256 x 256 tile, 8 waves as 2(M) x 4(N), a wave owns 128 x 64 = 4 x 2 MFMA tiles of 32 x 32 (128 accumulator
//     registers).  A 16x16x128 fragment set for a 128 x 80 wave tile would need 104 operand registers on top of 160
//     accumulators; 32x32x64 fragments cover twice the rows per register.
//   * K-step = 128 bytes = two 32x32x64 MFMAs deep; LDS rows are 128 B, a slot is (256 + 256) x 128 B = 64 KiB, two slots.
//     Per wave and K-step: 8 DMA pieces (8 rows x 128 B each), 24 ds_read_b128, 16 MFMAs x 64 cycles.
//   * swizzle: physical 16-byte chunk = chunk ^ ((row >> 1) & 7): the 16 lanes of a ds_read_b128 group (32 different rows
//     of one chunk column) then cover the 16 chunk positions of the 256-byte bank row exactly once.
// Lane maps (checked with exact data, tools/fp8_probe.hip): lane l holds row/col l & 31 and k = 32 (l >> 5) + j of the
// 64-deep MFMA step in its 32 operand bytes; C/D as every 32x32 MFMA: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
// Operand roles are swapped as in gemm16_v5.h (W fragment as the A operand), so a lane holds 4 consecutive n of one row m.
#pragma once
#include "gemm16_v5.h"
#include "misc_kernels.h"

#ifndef WM_GEMM8_GELU
#define WM_GEMM8_GELU gelu_e4m3_fast4      // dev A/B: -DWM_GEMM8_GELU=gelu_erf_fast4
#endif

namespace wm {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Gemm8Args {
    const unsigned char* A;       // [M][K] e4m3
    const unsigned char* W;       // [N][K] e4m3
    const float* wscale;          // [N]
    const float* bias;            // [N] or null
    const float* residual;        // [M][N] fp32 or null (may alias out32)
    float* out32;                 // [M][N] or null
    u16* out16;                   // [M][N] 16-bit (type T) or null
    unsigned char* out8;          // [M][N] e4m3 or null
    int M, N, K, act;
    unsigned long long* dbg;      // DBG instance only: [grid][4] stamps
    // PLANES instance (round 4): the residual stream as two 16-bit planes of rows, hi = T(x), lo = fp16(x - hi) (gemm16_v5.h "Split
    // stream"; rows here because the only reader of hi is the LayerNorm-to-e4m3 pass): in (res_hi, res_lo), out (out_hi, out_lo),
    // may alias.  The epilogue moves 8 bytes per element either way; the LayerNorm pass behind it reads 2 instead of 4.
    // Within a row every 256-column block is stored in the epilogue's PASS order (wm::plane_pos, wm_common.h): a pass reads and
    // writes one 256-byte run per row and plane (strict row-major gave 64-byte pieces and cost 31 us per launch).
    const u16* res_hi;
    const u16* res_lo;
    u16* out_hi;
    u16* out_lo;
};

// BKB = bytes (= fp8 elements) of K per LDS step: 128 (two MFMAs deep, 2 slots of 64 KiB, DMA issued after X_s) or
// 64 (one MFMA deep, 4 slots of 32 KiB, three K-steps of DMA in flight, each wave's pieces issued between the MFMAs of its
// own MFMA interval).  Measured (tools/gemm8_bench.py, B = 16): see DESIGN.md section 5.
template <int BKB_> struct G8 {
    static constexpr int BM = 256, BN = 256, BKB = BKB_, NSLOT = BKB_ == 128 ? 2 : 4;
    static constexpr int A_BYTES = BM * BKB, W_BYTES = BN * BKB, STAGE = A_BYTES + W_BYTES;
    static constexpr int LDS = NSLOT * STAGE + 32 * 1024;          // ring + epilogue room (second residual landing buffer)
    static constexpr int MT = 4, NT = 2, KS = BKB / 64;
    static constexpr int PROWS = 1024 / BKB;                       // rows per 1-KiB DMA piece
    static constexpr int PW = (BM / PROWS) / 8;                    // A (and W) pieces per wave and K-step
    static constexpr int AHEAD = NSLOT - 1;
};
constexpr int G8_BM = 256, G8_BN = 256;

// DBG (dev, WM_GEMM8_DBG=1): every workgroup records wall-clock stamps (entry, first barrier passed, loop end, stores
// acknowledged) into p.dbg; a separate instance, the product kernel carries none of it.
template <class T, int BKB, bool DBG = false, bool PLANES = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(Gemm8Args p) {
    using C = G8<BKB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long wt0 = 0, wt1 = 0, wt2 = 0, mt1 = 0, mt2 = 0;
    if constexpr (DBG) wt0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r32 = lane & 31, h = lane >> 5;
    const int K = p.K, ns = K / C::BKB;
    const char* Ab = (const char*)p.A;
    const char* Wb = (const char*)p.W;

    int m0, n0;
    {
        const int tilesM = p.M / C::BM, tilesN = p.N / C::BN;
        const int t = xcd_remap(blockIdx.x, gridDim.x);
        const int per_group = G16_GROUP_M * tilesN;
        const int group = t / per_group;
        const int first_m = group * G16_GROUP_M;
        const int gsz = min(G16_GROUP_M, tilesM - first_m);
        const int in_group = t - group * per_group;
        m0 = (first_m + in_group % gsz) * C::BM;
        n0 = (in_group / gsz) * C::BN;
    }

    // DMA piece = 1 KiB = PROWS rows of BKB bytes; lane -> row, physical 16-byte chunk.  Swizzle key of a row:
    //   BKB 128: (row >> 1) & 7 = (4 (piece & 1) + (lane >> 4)) & 7: even and odd pieces have their own per-lane source offset
    //   BKB  64: (row >> 2) & 3 = (lane >> 4) & 3, the same for every piece
    const size_t row_bytes = (size_t)K;
    unsigned lane_off[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        if constexpr (BKB == 128) lane_off[par] = (unsigned)(lane >> 3) * (unsigned)K + (unsigned)((((lane & 7) ^ ((4 * par + (lane >> 4)) & 7))) << 4);
        else lane_off[par] = (unsigned)(lane >> 2) * (unsigned)K + (unsigned)((((lane & 3) ^ ((lane >> 4) & 3))) << 4);
    }
    const char* a_wave = Ab + (size_t)(m0 + wave * 32) * row_bytes;             // this wave's A pieces: rows 32 wave .. + 31
    const char* w_wave = Wb + (size_t)(n0 + wave * 32) * row_bytes;
    auto piece_dma = [&](int slot, int s, int i) {                              // i < PW: A pieces, else W pieces
        const int j = i < C::PW ? i : i - C::PW;
        const char* base = (i < C::PW ? a_wave : w_wave) + (size_t)(j * C::PROWS) * row_bytes + (size_t)s * C::BKB;
        __builtin_amdgcn_global_load_lds(base + lane_off[j & 1],
                                         WM_LDS_PTR(smem + slot * C::STAGE + (i < C::PW ? 0 : C::A_BYTES) + (wave * C::PW + j) * 1024), 16, 0, 0);
    };
    auto stage = [&](int slot, int s) {
#pragma unroll
        for (int i = 0; i < 2 * C::PW; ++i) piece_dma(slot, s, i);
    };

    // fragment read: row r32 of a 32-row tile, 32 bytes = chunks c, c + 1 -> two ds_read_b128 whose addresses differ by XOR
    // constants only (the swizzle key has no bit in common with them)
    const int key = BKB == 128 ? ((r32 >> 1) & 7) : ((r32 >> 2) & 3);
    const int frag_off = r32 * C::BKB + ((((2 * h) ^ key)) << 4);
    const int rd_a = (wr * 128) * C::BKB + frag_off;
    const int rd_w = C::A_BYTES + (wc * 64) * C::BKB + frag_off;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    i32x8 af[C::KS][C::MT], wf[C::KS][C::NT];

    auto rd32 = [&](const char* base, int off) {
        const i32x4 lo = *(const i32x4*)(base + off), hi = *(const i32x4*)(base + (off ^ 16));
        return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto read_frags = [&](int slot) {
        const char* sS = smem + slot * C::STAGE;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
#pragma unroll
            for (int i = 0; i < C::NT; ++i) wf[ks][i] = rd32(sS, (rd_w + i * 32 * C::BKB) ^ (ks * 64));
#pragma unroll
            for (int i = 0; i < C::MT; ++i) af[ks][i] = rd32(sS, (rd_a + i * 32 * C::BKB) ^ (ks * 64));
        }
    };
    const int one = 0x7f7f7f7f;                              // E8M0 127 = 2^0 for every 32-element block
    // the MFMAs of one K-step; DMA: this wave's pieces of step `s` between the first of them (one piece per 64-cycle MFMA:
    // its issue hides behind the matrix pipe instead of lengthening the load interval)
    auto mfmas = [&](int slot, int s, auto dma_tag) {
        constexpr bool DMA = decltype(dma_tag)::value;
        int idx = 0;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
            for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0, one, 0, one);
                    if constexpr (DMA) {
                        if (idx < 2 * C::PW) piece_dma(slot, s, idx);
                    }
                    ++idx;
                }
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < 2 * C::PW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, C::KS * C::MT * C::NT - 2 * C::PW, 0);
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using NO = std::false_type;
    using YES = std::true_type;
    // DBG: waves 0 and 4 of workgroup 0 record s_memtime at 6 points of K-steps 4..7 (lane = (step - 4) * 6 + point)
    unsigned tmark = 0;
    auto mark = [&](int s, int k) {
        if constexpr (DBG) {
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            const int idx = (s - 4) * 6 + k;
            tmark = (lane == idx) ? t : tmark;
        }
    };

    if constexpr (BKB == 128) {
        // ---- two slots, one K-step of DMA in flight.  Both wave groups issue the pieces of step s + 1 right after X_s: slot
        // (s + 1) & 1 was last read for step s - 1, by waves 0-3 before Y_(s-1) and by waves 4-7 between Y_(s-1) and X_s
        // (drained with lgkmcnt(0) before they arrive at X_s).  A wave waits for its own pieces of step s (vmcnt(0): nothing
        // younger is outstanding at that point) before X_s, and every read of slot s & 1 follows X_s.  (Placing the pieces in
        // each group's own load interval as gemm16_v5.h does measured 3-7 % slower; between the MFMAs needs more than the
        // 256 registers: 128 accumulators + 96 operand registers leave no room for the address temporaries.)
        stage(0, 0);
        if (wr == 0) {
#pragma unroll 1
            for (int s = 0; s < ns; ++s) {
                wait_vmcnt<0>();
                barrier();                                  // X_s
                if constexpr (DBG) { if (s == 0) { wt1 = wall_clock64(); mt1 = __builtin_readcyclecounter(); } }
                if (s + 1 < ns) stage((s + 1) & 1, s + 1);
                read_frags(s & 1);
                barrier();                                  // Y_s
                mfmas(0, 0, NO{});
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < ns; ++s) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of the slot about to be overwritten are back
                wait_vmcnt<0>();
                barrier();                                  // X_s
                if (s + 1 < ns) stage((s + 1) & 1, s + 1);
                if (s > 0) mfmas(0, 0, NO{});               // step s - 1
                barrier();                                  // Y_s
                read_frags(s & 1);
            }
            mfmas(0, 0, NO{});                              // step ns - 1
        }
    } else {
        // ---- NSLOT slots, AHEAD = NSLOT - 1 K-steps of DMA in flight.  During step s a wave issues its pieces of step s + AHEAD
        // into slot (s - 1) % NSLOT, between the MFMAs of its MFMA interval (waves 0-3: after Y_s; waves 4-7: after X_s).  That
        // slot was last read for step s - 1: by waves 0-3 before Y_(s-1), by waves 4-7 between Y_(s-1) and X_s (drained with
        // lgkmcnt(0) before X_s), so every issue follows the last read.  A wave waits for its own pieces of step s -- all but
        // the (AHEAD - 1) younger steps' pieces -- before X_s; every read of slot s % NSLOT follows X_s.
        constexpr int PWS = 2 * C::PW;                      // pieces per wave and step
        static_assert(C::AHEAD == 3, "wait ladder below is written for three steps in flight");
        auto wait_tail = [&](int s) {                       // last AHEAD steps: fewer younger pieces outstanding
            if (s + 2 < ns) wait_vmcnt<2 * PWS>();
            else if (s + 1 < ns) wait_vmcnt<PWS>();
            else wait_vmcnt<0>();
        };
#pragma unroll
        for (int i = 0; i < C::AHEAD; ++i) stage(i, i);     // ns > AHEAD (K >= 256)
        int slot = 0;                                       // slot of step s
        auto prev = [](int v) { return v == 0 ? C::NSLOT - 1 : v - 1; };
        auto next = [](int v) { return v == C::NSLOT - 1 ? 0 : v + 1; };
        const int n_main = ns - C::AHEAD;                   // steps that still have a step s + AHEAD to request
        // (the loops are peeled so that each has ONE MFMA block: with / without the DMA pieces between the MFMAs)
        if (wr == 0) {
#pragma unroll 1
            for (int s = 0; s < n_main; ++s) {
                mark(s, 0);
                wait_vmcnt<2 * PWS>();
                mark(s, 1);
                barrier();                                  // X_s
                if constexpr (DBG) { if (s == 0) { wt1 = wall_clock64(); mt1 = __builtin_readcyclecounter(); } }
                mark(s, 2);
                read_frags(slot);
                mark(s, 3);
                barrier();                                  // Y_s
                mark(s, 4);
                mfmas(prev(slot), s + C::AHEAD, YES{});
                mark(s, 5);
                slot = next(slot);
            }
#pragma unroll 1
            for (int s = n_main; s < ns; ++s) {
                wait_tail(s);
                barrier();                                  // X_s
                read_frags(slot);
                barrier();                                  // Y_s
                mfmas(0, 0, NO{});
                slot = next(slot);
            }
        } else {
            {                                               // s = 0: nothing to multiply yet
                wait_vmcnt<2 * PWS>();
                barrier();                                  // X_0
                stage(prev(slot), C::AHEAD);
                barrier();                                  // Y_0
                read_frags(slot);
                slot = next(slot);
            }
#pragma unroll 1
            for (int s = 1; s < n_main; ++s) {
                mark(s, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_vmcnt<2 * PWS>();
                mark(s, 1);
                barrier();                                  // X_s
                mark(s, 2);
                mfmas(prev(slot), s + C::AHEAD, YES{});      // MFMAs of step s - 1
                mark(s, 3);
                barrier();                                  // Y_s
                mark(s, 4);
                read_frags(slot);
                mark(s, 5);
                slot = next(slot);
            }
#pragma unroll 1
            for (int s = n_main > 1 ? n_main : 1; s < ns; ++s) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_tail(s);
                barrier();                                  // X_s
                mfmas(0, 0, NO{});                          // MFMAs of step s - 1
                barrier();                                  // Y_s
                read_frags(slot);
                slot = next(slot);
            }
            mfmas(0, 0, NO{});                              // step ns - 1
        }
    }

    if constexpr (DBG) {
        wt2 = wall_clock64(); mt2 = __builtin_readcyclecounter();
        if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane < 24) ((unsigned*)(p.dbg + (size_t)gridDim.x * 4 + 2))[wr * 24 + lane] = tmark;
    }
    // ---- epilogue (through LDS so that every global access is row-contiguous; see gemm16_v5.h) ----
    // lane holds, per 32 x 32 tile (mi, ni): row m = r32, columns n = 8 g + 4 h + (0..3) for g = 0..3 (registers 4 g .. 4 g + 3)
    // per-channel weight scales and biases of this lane's 8 column groups: requested before the barrier, so their round
    // trip overlaps the last MFMAs draining
    f32x4 scv[C::NT][4], biv[C::NT][4];
#pragma unroll
    for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = 64 * wc + 32 * ni + 8 * g + 4 * h;
            scv[ni][g] = *(const f32x4*)(p.wscale + n0 + nl);
            biv[ni][g] = p.bias ? *(const f32x4*)(p.bias + n0 + nl) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
    barrier();                                              // every wave is done with the ring
    auto scaled = [&](int mi, int ni, int g, const f32x4& sc, const f32x4& bi) {
        f32x4 v{acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], bi[j]);
        return v;
    };
    const int act = p.act & 0xff;
    if (PLANES || p.residual != nullptr) {
        // fp32 + residual (proj, lin2): 8 passes (mi, ni) of 64 rows x 128 columns (4 strips of 32, one per wc); the residual
        // strip comes in by LDS-DMA one pass ahead into two landing buffers; then 16-byte chunks along the rows
        constexpr int ROWB = 128 * 4 + 16, STG = 0, L0 = 40 * 1024, L1 = 80 * 1024, LAND = 64 * 128 * 4;
        static_assert(64 * ROWB <= L0 && L0 + LAND <= L1 && L1 + LAND <= C::LDS, "epilogue LDS map");
        auto col_of = [&](int ch, int ni) { return n0 + 64 * (ch >> 3) + 32 * ni + 4 * (ch & 7); };
        auto res_dma = [&](int q) {
            const int mi = q >> 1, ni = q & 1;
            char* dst = smem + ((q & 1) ? L1 : L0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = wave * 4 + i;
                if constexpr (PLANES) {
                    // landing image [plane][64 rows][4 strips x 64 B]: pieces 0..15 = hi, 16..31 = lo; a lane's 16 B = 8 columns of one strip.
                    // The planes keep each 256-column block in PASS order (plane_pos below): the 4 x 32 columns a pass touches are one
                    // 256-byte run of the row
                    const int cidx = (piece & 15) * 64 + lane, rr = cidx >> 4, j = cidx & 15;
                    const int m = m0 + (rr >> 5) * 128 + mi * 32 + (rr & 31);
                    const u16* src = (piece >> 4 ? p.res_lo : p.res_hi) + (size_t)m * p.N + n0 + 128 * ni + j * 8;
                    __builtin_amdgcn_global_load_lds((const char*)src, WM_LDS_PTR(dst + piece * 1024), 16, 0, 0);
                } else {
                const int rr = piece * 2 + (lane >> 5), ch = lane & 31;
                const int m = m0 + (rr >> 5) * 128 + mi * 32 + (rr & 31);
                __builtin_amdgcn_global_load_lds((const char*)(p.residual + (size_t)m * p.N + col_of(ch, ni)), WM_LDS_PTR(dst + piece * 1024), 16, 0, 0);
                }
            }
        };
        res_dma(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int mi = q >> 1, ni = q & 1;
            if (q + 1 < 8) res_dma(q + 1);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(smem + STG + (wr * 32 + r32) * ROWB + (wc * 32 + 8 * g + 4 * h) * 4) = scaled(mi, ni, g, scv[ni][g], biv[ni][g]);
            if (q + 1 < 8) wait_vmcnt<4>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            barrier();
            const char* land = smem + ((q & 1) ? L1 : L0);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int c = it * 512 + tid, rr = c >> 5, ch = c & 31;
                const int m = m0 + (rr >> 5) * 128 + mi * 32 + (rr & 31);
                f32x4 res;
                if constexpr (PLANES) {
                    const int off = rr * 256 + ch * 8;
                    const typename T::vec4 h4 = *(const typename T::vec4*)(land + off);
                    const f16x4 l4 = *(const f16x4*)(land + 16384 + off);
#pragma unroll
                    for (int j = 0; j < 4; ++j) res[j] = T::to_f32(h4[j]) + (float)l4[j];
                } else {
                    res = *(const f32x4*)(land + c * 16);
                }
                const f32x4 v = *(const f32x4*)(smem + STG + rr * ROWB + ch * 16) + res;
                if constexpr (PLANES) {
                    // (8 columns per thread with 16-byte plane accesses measured 0.7 ms per step slower than this: profiles/r4_dev)
                    typename T::vec4 o;
                    f16x4 lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = T::from_f32(v[j]);
                        lo[j] = FP16::from_f32(v[j] - T::to_f32(o[j]));
                    }
                    *(typename T::vec4*)(p.out_hi + (size_t)m * p.N + n0 + 128 * ni + ch * 4) = o;
                    *(f16x4*)(p.out_lo + (size_t)m * p.N + n0 + 128 * ni + ch * 4) = lo;
                    continue;
                }
                if (p.out32) *(f32x4*)(p.out32 + (size_t)m * p.N + col_of(ch, ni)) = v;
                if (p.out16) {
                    typename T::vec4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                    *(typename T::vec4*)(p.out16 + (size_t)m * p.N + col_of(ch, ni)) = o;
                }
            }
            if (q + 1 < 8) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                barrier();
            }
        }
    } else if (p.out8 != nullptr) {
        // e4m3 output (lin1 + GELU): one pass, 256 rows x 256 bytes
        constexpr int ROWB = 256 + 16;
        static_assert(256 * ROWB <= C::LDS, "epilogue LDS map");
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 64 * wc + 32 * ni + 8 * g + 4 * h;
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi) {
                    f32x4 v = scaled(mi, ni, g, scv[ni][g], biv[ni][g]);
                    if (act == ACT_GELU) v = WM_GEMM8_GELU(v);       // e4m3 output: the 3-bit-mantissa-grade form (wm_common.h)
                    else if (act == ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    *(unsigned*)(smem + (wr * 128 + mi * 32 + r32) * ROWB + nl) = pack4_e4m3(v);
                }
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int c = it * 512 + tid, rr = c >> 4, ch = c & 15;
            const f32x4 v = *(const f32x4*)(smem + rr * ROWB + ch * 16);
            *(f32x4*)(p.out8 + (size_t)(m0 + rr) * p.N + n0 + ch * 16) = v;
        }
    } else {
        // 16-bit output (qkv): one pass, 256 rows x 256 columns
        constexpr int ROWB = 256 * 2 + 16;
        static_assert(256 * ROWB <= C::LDS, "epilogue LDS map");
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 64 * wc + 32 * ni + 8 * g + 4 * h;
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi) {
                    f32x4 v = scaled(mi, ni, g, scv[ni][g], biv[ni][g]);
                    if (act == ACT_GELU) v = gelu_erf_fast4(v);
                    else if (act == ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    typename T::vec4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                    *(typename T::vec4*)(smem + (wr * 128 + mi * 32 + r32) * ROWB + nl * 2) = o;
                }
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int c = it * 512 + tid, rr = c >> 5, ch = c & 31;
            const f32x4 v = *(const f32x4*)(smem + rr * ROWB + ch * 16);
            *(f32x4*)((char*)p.out16 + ((size_t)(m0 + rr) * p.N + n0) * 2 + ch * 16) = v;
        }
    }
    if constexpr (DBG) {
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long wt3 = wall_clock64();
            if (lane == 0) {
                unsigned long long* r = p.dbg + (size_t)blockIdx.x * 4;
                r[0] = wt0; r[1] = wt1; r[2] = wt2; r[3] = wt3;
                if (blockIdx.x == 0) { r[0] = wt0; p.dbg[(size_t)gridDim.x * 4] = mt2 - mt1; p.dbg[(size_t)gridDim.x * 4 + 1] = wt2 - wt1; }
            }
        }
    }
}

// 16-bit [n] -> e4m3 [n], unit scale, saturating (attention output -> the A operand of proj)
template <class T>
__global__ __launch_bounds__(256) void cvt_16_to_fp8_kernel(const u16* __restrict__ in, unsigned char* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const typename T::vec8 v = *(const typename T::vec8*)(in + i * 8);
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = T::to_f32(v[j]); b[j] = T::to_f32(v[4 + j]); }
        uint2 o;
        o.x = pack4_e4m3(a);
        o.y = pack4_e4m3(b);
        *(uint2*)(out + i * 8) = o;
    }
}

// fp32 [n] -> e4m3 [n] (tests)
__global__ __launch_bounds__(256) void cvt_f32_to_fp8_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        *(unsigned*)(out + i * 4) = pack4_e4m3(*(const f32x4*)(in + i * 4));
}

}  // namespace wm
