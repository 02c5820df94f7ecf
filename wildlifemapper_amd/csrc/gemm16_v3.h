// 16-bit MFMA GEMM, 256 x BN tile with 128 x (BN/WN) per wave, K-steps of 32 (gfx950).
//
// Measurements on gemm16_v2.h (256x160x64, 64x80 per wave) showed the block GEMMs are bound by the rate at
// which a CU can pull operand tiles from L2 into LDS (~48 GB/s per CU against a ~70 GB/s ceiling for
// L2-served LDS fills, MI355X_MICROARCH.md "Indexed rows") and, per tile, by the C write: neither persistent
// workgroups, nor row-contiguous stores, nor two workgroups per CU moved it.  What does is fewer operand
// bytes per FLOP.  This kernel doubles the per-wave tile to 128 x 80 (160 accumulator registers) and, with
// 8 waves as 2(M) x 4(N), the workgroup tile to 256 x 320:
//      operand bytes per FLOP   256x160: 1/98     256x320: 1/142   (-31 %)
//      fragment reads per MFMA  0.45 -> 0.325
// and N = 1280 / 3840 / 5120 at M = 16384 (4 tiles per GPU) still give 256 / 768 / 1024 workgroups: whole
// rounds of the 256 CUs.  An XCD's 32 co-resident tiles form an 8(M) x 4(N) block of the grouped order, i.e.
// 2048 x 1280 of C sharing 8 A panels and 4 W panels in one L2.
//
//   template <BN, WN>: waves = 2 x WN, per-wave columns BN / WN (80 or 64):
//      <320,4> block GEMMs          <256,4> N = 256 / 1024 / 2048 (neck, HFC adaptor)
//      <160,2> / <128,2>            4-wave variants (two workgroups per CU), kept for A/B runs
//   K-step 32: LDS rows are 64 B; 3-slot ring of (256 + BN) x 64 B (108 KiB at BN = 320).
//   DMA piece = 16 rows x 64 B; swizzle phys_chunk = chunk ^ ((-(row >> 2)) & 3) on the SOURCE address and
//   on the ds_read_b128 address (conflict-free lane groups).
//   AMODE 2 (round 4; im2col-free 16x16 / stride-16 patch embeds, image_encoder.py:386-450): the A tile is gathered from the
//   16-bit NCHW image, see dma_a.
//   AMODE 1 (im2col-free 3x3 conv, the neck's second conv image_encoder.py:113-119): the A tile of K-step s
//   is the 32-channel slice ci0 = (32 s) % C of tap (32 s) / C of the NHWC activation, shifted by the tap's
//   (dy, dx); the DMA source address is computed per lane and points at a zero page outside the image.
//   W pieces do not divide evenly over the waves: waves < W_REM issue one more, and wait with their own
//   counted vmcnt.  Synchronisation as gemm16_v2.h (counted vmcnt + raw s_barrier, DMA of step s+2 after the
//   barrier, spread between the MFMAs).
#pragma once
#include <type_traits>

#include "gemm16.h"

#ifndef WM_GEMM_TIMING_BITS
#define WM_GEMM_TIMING_BITS 0
#endif

namespace wm {

template <int BN, int WN> struct G3 {
    static constexpr int BM = 256, BK = 32, WAVES = 2 * WN, THREADS = 64 * WAVES;
    static constexpr int WCOLS = BN / WN, NT = WCOLS / 16, MT = 8;
    static constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64;
    static constexpr int STAGE = A_BYTES + W_BYTES;
    static constexpr int LDS = 3 * STAGE;
    static constexpr int A_PIECES = (BM / 16) / WAVES;               // per wave
    static constexpr int W_TOTAL = BN / 16, W_LO = W_TOTAL / WAVES, W_REM = W_TOTAL % WAVES;
    static constexpr int P_LO = A_PIECES + W_LO;                     // pieces per step of waves >= W_REM
    static_assert((BM / 16) % WAVES == 0 && WCOLS % 16 == 0, "tile shape");
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 12, "extend the table");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

template <class T, int BN, int WN, int AMODE = 0>
__global__ __launch_bounds__((G3<BN, WN>::THREADS), 2) void gemm16v3_kernel(Gemm16Args p) {
    using C = G3<BN, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = p.K, ns = K / C::BK;
    const char* Ab = (const char*)p.A;
    const char* Wb = (const char*)p.W;
    const bool extra = wave < C::W_REM;                              // this wave issues W_LO + 1 W pieces
    constexpr bool TB = WM_GEMM_TIMING_BITS != 0;                        // timing experiments (tools/gemm_bench.py), off in the product
    const bool dbg_nostore = TB && (p.act & 0x100) != 0, dbg_nodma = TB && (p.act & 0x200) != 0;
    const int act = p.act & 0xff;

    // grouped tile order + XCD remap (as gemm16_v2.h)
    int m0, n0;
    {
        const int tilesM = p.M / C::BM, tilesN = p.N / BN;
        const int t = xcd_remap(blockIdx.x, gridDim.x);
        const int per_group = G16_GROUP_M * tilesN;
        const int group = t / per_group;
        const int first_m = group * G16_GROUP_M;
        const int gsz = min(G16_GROUP_M, tilesM - first_m);
        const int in_group = t - group * per_group;
        m0 = (first_m + in_group % gsz) * C::BM;
        n0 = (in_group / gsz) * BN;
    }

    // DMA: per-lane part of the source address (row in piece, swizzled chunk); piece bases are wave-uniform
    const unsigned lane_off = (unsigned)(lane >> 2) * (unsigned)(K * 2) + (unsigned)((((lane & 3) ^ ((0 - (lane >> 4)) & 3))) << 4);
    const size_t row_bytes = (size_t)K * 2;
    auto dma_a = [&](int slot, int s, int i) {
        const int seg = wave * C::A_PIECES + i;
        if constexpr (AMODE == 0) {
            const char* base = Ab + (size_t)(m0 + seg * 16) * row_bytes + (size_t)s * 64;
            if (dbg_nodma) base = Ab;
            __builtin_amdgcn_global_load_lds(base + (dbg_nodma ? (lane_off & 1023u) : lane_off), WM_LDS_PTR(smem + slot * C::STAGE + seg * 1024), 16, 0, 0);
        } else if constexpr (AMODE == 2) {
            // 16 x 16 / stride 16 patch embed without an im2col buffer (image_encoder.py:386-450): A[m][k] with m = (b, py, px) and
            // k = c * 256 + ky * 16 + kx is read straight from the 16-bit NCHW image [B][conv_c][1024][1024].  K-step s covers
            // channel s >> 3, image rows 2 (s & 7) and + 1 of the patch, all 16 kx: the lane's 16-byte chunk `ch` of the 64-byte
            // LDS row is (ky = 2 (s & 7) + (ch >> 1), kx = 8 (ch & 1) .. + 7).  A piece's 16 rows are 16 consecutive patches of one
            // patch row, so it reads two 512-byte runs of the image.
            const int m = m0 + seg * 16 + (lane >> 2);
            const int pix = m & 4095, py = pix >> 6, px = pix & 63, b = m >> 12;
            const int ch = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
            const size_t row = ((size_t)b * p.conv_c + (s >> 3)) * 1024 + (size_t)(py * 16 + 2 * (s & 7) + (ch >> 1));
            const char* src = Ab + (row * 1024 + (size_t)(px * 16 + (ch & 1) * 8)) * 2;
            __builtin_amdgcn_global_load_lds(src, WM_LDS_PTR(smem + slot * C::STAGE + seg * 1024), 16, 0, 0);
        } else {
            // pixel of this lane's row; 16 consecutive rows of a piece lie in one image row (64 % 16 == 0)
            const int m = m0 + seg * 16 + (lane >> 2);
            const int pix = m & 4095, y = pix >> 6, x = pix & 63;
            const int kg = s * 32, tap = kg / p.conv_c, ci0 = kg - tap * p.conv_c;     // wave-uniform
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int yy = y + dy, xx = x + dx;
            const unsigned chunk_off = (unsigned)((((lane & 3) ^ ((0 - (lane >> 4)) & 3))) << 4);
            const bool inside = (unsigned)yy < 64u && (unsigned)xx < 64u;
            const char* src = inside ? Ab + ((size_t)(m + dy * 64 + dx) * p.conv_c + ci0) * 2 + chunk_off
                                     : (const char*)p.zero_page + chunk_off;
            __builtin_amdgcn_global_load_lds(src, WM_LDS_PTR(smem + slot * C::STAGE + seg * 1024), 16, 0, 0);
        }
    };
    auto dma_w = [&](int slot, int s, int seg) {
        const char* base = Wb + (size_t)(n0 + seg * 16) * row_bytes + (size_t)s * 64;
        if (dbg_nodma) base = Wb;
        __builtin_amdgcn_global_load_lds(base + (dbg_nodma ? (lane_off & 1023u) : lane_off), WM_LDS_PTR(smem + slot * C::STAGE + C::A_BYTES + seg * 1024), 16, 0, 0);
    };
    // pieces every wave issues (P_LO of them), then the remainder piece of the first W_REM waves
    auto stage_uniform = [&](int slot, int s) {
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) dma_a(slot, s, i);
#pragma unroll
        for (int i = 0; i < C::W_LO; ++i) dma_w(slot, s, wave * C::W_LO + i);
    };
    auto stage_extra = [&](int slot, int s) {
        if constexpr (C::W_REM > 0) {
            if (extra) dma_w(slot, s, C::WAVES * C::W_LO + wave);
        }
    };

    // fragment reads: row = tile*16 + fr, logical chunk fq (k = 8 fq .. 8 fq + 7)
    const int frag_off = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);
    const int rd_a = (wr * 128) * 64 + frag_off;
    const int rd_w = C::A_BYTES + (wc * C::WCOLS) * 64 + frag_off;

    f32x4 acc[C::MT][C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](int slot, int dma_s, auto dma_tag, auto last_tag) {
        constexpr bool DMA = decltype(dma_tag)::value, LAST = decltype(last_tag)::value;
        if constexpr (LAST) {
            wait_vmcnt<0>();
        } else if constexpr (C::W_REM > 0) {
            if (extra) wait_vmcnt<C::P_LO + 1>(); else wait_vmcnt<C::P_LO>();
        } else {
            wait_vmcnt<C::P_LO>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int nsl = slot + 2;
        if (nsl >= 3) nsl -= 3;
        if constexpr (DMA) stage_extra(nsl, dma_s);                  // own basic block, ahead of the scheduled region
        const char* sS = smem + slot * C::STAGE;
        typename T::vec8 wf[C::NT], af[C::MT];
        // issue order A0, W0..W(NT-1), A1..: the first MFMA needs only the first two reads back
        af[0] = *(const typename T::vec8*)(sS + rd_a);
#pragma unroll
        for (int i = 0; i < C::NT; ++i) wf[i] = *(const typename T::vec8*)(sS + rd_w + i * 1024);
#pragma unroll
        for (int i = 1; i < C::MT; ++i) af[i] = *(const typename T::vec8*)(sS + rd_a + i * 1024);
        if constexpr (DMA) stage_uniform(nsl, dma_s);
#pragma unroll
        for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) acc[mi][ni] = T::mfma16(wf[ni], af[mi], acc[mi][ni]);
        // fragment reads first, then MFMAs with one DMA piece after every NT of them
        constexpr int NM = C::MT * C::NT;
        __builtin_amdgcn_sched_group_barrier(0x100, C::NT + C::MT, 0);
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < C::P_LO; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, C::NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - C::NT * C::P_LO, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        }
    };

    stage_uniform(0, 0);
    stage_extra(0, 0);
    if (ns > 1) { stage_uniform(1, 1); stage_extra(1, 1); }
    int slot = 0, s = 0;
    for (; s + 2 < ns; ++s) {
        kstep(slot, s + 2, std::true_type{}, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (s + 1 < ns) {
        kstep(slot, 0, std::false_type{}, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
    }
    kstep(slot, 0, std::false_type{}, std::true_type{});

    // epilogue: lane holds C[m][n..n+3]
    const int res_mod = p.res_mod > 0 ? p.res_mod : p.M;
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
        const int m = m0 + wr * 128 + mi * 16 + fr;
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            const int n = n0 + wc * C::WCOLS + ni * 16 + fq * 4;
            f32x4 v = acc[mi][ni];
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (act == ACT_GELU) {
                v = gelu_erf_fast4(v);      // the same arithmetic in every GEMM kernel: a tile's bits must not depend on which one its batch size selects
            } else if (act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (p.residual) v += *(const f32x4*)(p.residual + (size_t)(m % res_mod) * p.N + n);
            if (dbg_nostore) { if (v[0] == 12345.678f) p.out16[0] = 1; continue; }
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
