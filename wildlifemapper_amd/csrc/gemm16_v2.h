// 16-bit MFMA GEMM, 256 x BN x 64 tile, 3-deep LDS-DMA ring (gfx950).
//
// Same contract and epilogue as gemm16.h, for M % 256 == 0 and N % BN == 0 with
// BN = 128 or 160.  BN = 160 exists for the block GEMMs at 4 tiles per GPU
// (M = 16384): N = 1280 / 3840 / 5120 give 512 / 1536 / 2048 workgroups, whole
// multiples of the 256 CUs at one workgroup per CU, where 128- or 256-wide tiles
// leave the last round 25-50 % empty.
//
// 8 waves as 4(M) x 2(N); a wave owns 64 x BN/2 of C (4 x NT MFMA 16x16x32 tiles).
// K-tiles of 64 go through a 3-slot LDS ring filled by global_load_lds (16 B per
// lane, XOR swizzle on the SOURCE address + on the ds_read address).  Two K-tiles
// are in flight while the third is consumed: the loop waits with a COUNTED
// s_waitcnt vmcnt(PIECES) -- never 0 -- then a raw s_barrier, so the DMA issued
// for tile kt+2 stays in flight across the barrier
// (cdna_hip_programming.md §5 "Pipelining across barriers").
//   RAW: every wave waits for its own pieces of tile kt (vmcnt) before the barrier,
//        every reader passes the barrier before its first ds_read of tile kt.
//   WAR: slot (kt+2)%3 was last read in iteration kt-1; those reads were consumed
//        by MFMAs that precede this iteration's barrier in every wave.
#pragma once
#include <type_traits>

#include "gemm16.h"

namespace wm {

template <int BN> struct G2 {
    static constexpr int BM = 256, BK = 64, NT = BN / 32;           // n-tiles per wave
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    static constexpr int STAGE = A_BYTES + W_BYTES;
    static constexpr int LDS = 3 * STAGE;
    static constexpr int A_PIECES = BM / 8 / 8;                      // 1-KiB pieces per wave: 4
    static constexpr int W_PIECES = (BN / 8 + 7) / 8;                // 2 (BN=128) or 3 (BN=160, 4 duplicates)
    static constexpr int PIECES = A_PIECES + W_PIECES;
};

template <class T, int BN>
__global__ __launch_bounds__(512, 2) void gemm16v2_kernel(Gemm16Args p) {
    using C = G2<BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tilesM = p.M / C::BM, tilesN = p.N / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = G16_GROUP_M * tilesN;
    const int group = lid / per_group;
    const int first_m = group * G16_GROUP_M;
    const int gsz = min(G16_GROUP_M, tilesM - first_m);
    const int in_group = lid - group * per_group;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int m0 = tm * C::BM, n0 = tn * BN;
    const int K = p.K, nk = K / C::BK;

    const char* Ab = (const char*)p.A;
    const char* Wb = (const char*)p.W;

    // DMA piece i of this wave: 8 rows x 128 B; lane -> (row in piece, swizzled source chunk)
    // fixed-size arrays: hipcc (ROCm 7.2) drops the host stub of a __global__ template whose lambda
    // captures an array of template-dependent size
    static_assert(C::A_PIECES == 4 && C::W_PIECES <= 3, "piece arrays");
    size_t a_off[4], w_off[3];
#pragma unroll
    for (int i = 0; i < C::A_PIECES; ++i) {
        const int seg = wave * C::A_PIECES + i;
        const int r = seg * 8 + (lane >> 3);
        a_off[i] = ((size_t)(m0 + r) * K) * 2 + (((lane & 7) ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < C::W_PIECES; ++i) {
        const int seg = (wave * C::W_PIECES + i) % (BN / 8);     // BN=160: pieces 20..23 re-load 0..3 (uniform count)
        const int r = seg * 8 + (lane >> 3);
        w_off[i] = ((size_t)(n0 + r) * K) * 2 + (((lane & 7) ^ (r & 7)) << 4);
    }

    auto stage = [&](int slot, int kt) {
        char* sA = smem + slot * C::STAGE;
        char* sW = sA + C::A_BYTES;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) {
            const int seg = wave * C::A_PIECES + i;
            __builtin_amdgcn_global_load_lds(Ab + a_off[i] + (size_t)kt * 128, WM_LDS_PTR(sA + seg * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::W_PIECES; ++i) {
            const int seg = (wave * C::W_PIECES + i) % (BN / 8);
            __builtin_amdgcn_global_load_lds(Wb + w_off[i] + (size_t)kt * 128, WM_LDS_PTR(sW + seg * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][C::NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    int a_rd[4][2], w_rd[C::NT][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_rd[i][ks] = ra * 128 + (((ks * 4 + fq) ^ (ra & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < C::NT; ++i) {
        const int rw = wc * (BN / 2) + i * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_rd[i][ks] = C::A_BYTES + rw * 128 + (((ks * 4 + fq) ^ (rw & 7)) << 4);
    }

    stage(0, 0);
    if (nk > 1) stage(1, 1);

    // One K-tile: [counted wait + barrier] then 2 x (4+NT) fragment reads and 2 x 4*NT MFMAs, with the
    // next-but-one tile's DMA pieces spread between the MFMAs (sched_group_barrier) so that LDS-DMA
    // issue (~100 cycles a piece) overlaps the matrix pipe instead of preceding it.
    auto ktile = [&](int slot, int kt, auto dma_tag, auto last_tag) {
        constexpr bool DMA = decltype(dma_tag)::value, LAST = decltype(last_tag)::value;
        if constexpr (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (C::PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // keep this tile's ds_reads below the barrier
        const char* sS = smem + slot * C::STAGE;
        typename T::vec8 af[2][4], wf[2][C::NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *(const typename T::vec8*)(sS + a_rd[i][ks]);
#pragma unroll
            for (int i = 0; i < C::NT; ++i) wf[ks][i] = *(const typename T::vec8*)(sS + w_rd[i][ks]);
        }
        if constexpr (DMA) {
            int ns = slot + 2;
            if (ns >= 3) ns -= 3;
            stage(ns, kt + 2);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) acc[mi][ni] = T::mfma16(wf[ks][ni], af[ks][mi], acc[mi][ni]);
        // schedule: ks0 fragments, then ks0 MFMAs interleaved with ks1 fragment reads, then ks1 MFMAs
        // interleaved with the DMA pieces
        constexpr int NF = 4 + C::NT, NM = 4 * C::NT;
        __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * NF, 0);
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < C::PIECES; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * C::PIECES, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        }
    };

    int slot = 0;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
        ktile(slot, kt, std::true_type{}, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (kt + 1 < nk) {
        ktile(slot, kt, std::false_type{}, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
        ++kt;
    }
    ktile(slot, kt, std::false_type{}, std::true_type{});

    const int res_mod = p.res_mod > 0 ? p.res_mod : p.M;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            const int n = n0 + wc * (BN / 2) + ni * 16 + fq * 4;
            f32x4 v = acc[mi][ni];
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf_fast(v[j]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (p.residual) v += *(const f32x4*)(p.residual + (size_t)(m % res_mod) * p.N + n);
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
