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
// Inside a K-tile the DMA pieces of tile kt+2 are spread between the MFMAs
// (sched_group_barrier): an LDS-DMA piece costs ~100 issue cycles, which would
// otherwise sit in front of the matrix work of both waves of a SIMD.
//
// Addressing: a DMA piece is 8 rows x 128 B.  Its per-lane part (row-in-piece and
// swizzled 16-B chunk) is the same for every piece, so one 32-bit lane offset is
// added to wave-uniform piece bases; fragment reads use one address per k-step plus
// immediate offsets (the swizzle term depends only on row & 7 = lane & 7).
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
    static constexpr int MIN_STORES = 4 * NT;                        // stores per lane of the leanest epilogue
};

// Per-wave machinery of the kernel.
template <class T, int BN> struct G2Core {
    using C = G2<BN>;
    char* smem;
    const char* Ab;
    const char* Wb;
    int K, wave, lane, wr, wc, fr, fq;
    unsigned lane_off;          // per-lane byte offset inside a DMA piece's source rows
    int rd_a[2], rd_w[2];       // fragment read offsets (bytes in a stage) for k-step 0 / 1, first m / n tile
    f32x4 acc[4][C::NT];

    __device__ __forceinline__ void init(char* smem_, const Gemm16Args& p) {
        smem = smem_;
        Ab = (const char*)p.A;
        Wb = (const char*)p.W;
        K = p.K;
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        wr = wave >> 1; wc = wave & 1;
        fr = lane & 15; fq = lane >> 4;
        const int rip = lane >> 3;                                   // row in piece == (row & 7)
        lane_off = (unsigned)rip * (unsigned)(K * 2) + (unsigned)(((lane & 7) ^ rip) << 4);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ((ks * 4 + fq) ^ (fr & 7)) << 4;          // rows of all m / n tiles share row & 7 = fr & 7
            rd_a[ks] = (wr * 64 + fr) * 128 + sw;
            rd_w[ks] = C::A_BYTES + (wc * (BN / 2) + fr) * 128 + sw;
        }
        zero();
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < C::NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // issue the DMA of K-tile kt of the output tile at (m0, n0) into ring slot `slot`
    __device__ __forceinline__ void stage(int slot, int m0, int n0, int kt) const {
        char* sA = smem + slot * C::STAGE;
        char* sW = sA + C::A_BYTES;
        const size_t row_bytes = (size_t)K * 2;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) {
            const int seg = wave * C::A_PIECES + i;
            const char* base = Ab + (size_t)(m0 + seg * 8) * row_bytes + (size_t)kt * 128;     // wave-uniform
            __builtin_amdgcn_global_load_lds(base + lane_off, WM_LDS_PTR(sA + seg * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::W_PIECES; ++i) {
            const int seg = (wave * C::W_PIECES + i) % (BN / 8);     // BN=160: pieces 20..23 re-load 0..3 (uniform count)
            const char* base = Wb + (size_t)(n0 + seg * 8) * row_bytes + (size_t)kt * 128;
            __builtin_amdgcn_global_load_lds(base + lane_off, WM_LDS_PTR(sW + seg * 1024), 16, 0, 0);
        }
    }
    // One K-tile out of ring slot `slot`.  WAIT selects the counted wait in front of the barrier:
    //   0: vmcnt(PIECES)   1: (extra_ok ? vmcnt(PIECES + MIN_STORES) : vmcnt(PIECES))   2: vmcnt(0)
    // DMA: also issue K-tile dma_kt of tile (dm0, dn0) into slot+2, spread between the MFMAs.
    template <bool DMA, int WAIT>
    __device__ __forceinline__ void ktile(int slot, int dm0, int dn0, int dma_kt, bool extra_ok) {
        if constexpr (WAIT == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (WAIT == 1 && extra_ok) {
                if constexpr (C::PIECES == 6) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
            } else {
                if constexpr (C::PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // keep this tile's ds_reads below the barrier
        const char* sS = smem + slot * C::STAGE;
        typename T::vec8 af[2][4], wf[2][C::NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* pa = sS + rd_a[ks];
            const char* pw = sS + rd_w[ks];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *(const typename T::vec8*)(pa + i * 2048);
#pragma unroll
            for (int i = 0; i < C::NT; ++i) wf[ks][i] = *(const typename T::vec8*)(pw + i * 2048);
        }
        if constexpr (DMA) {
            int ns = slot + 2;
            if (ns >= 3) ns -= 3;
            stage(ns, dm0, dn0, dma_kt);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni) acc[mi][ni] = T::mfma16(wf[ks][ni], af[ks][mi], acc[mi][ni]);
        // schedule: ks0 fragments; ks0 MFMAs interleaved with ks1 fragment reads; ks1 MFMAs interleaved with DMA pieces
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
    }
    // bias / activation / residual / stores straight from the MFMA layout: lane holds C[m][n..n+3].
    // (Measured: parking the tile in LDS to store contiguous rows, or keeping the ring full across tiles in a
    // persistent loop, change nothing -- at one workgroup per CU every CU writes its tile at the same moment
    // and the C write runs at the chip's HBM write rate; see DESIGN.md "GEMM".)
    __device__ __forceinline__ void epilogue(const Gemm16Args& p, int m0, int n0) const {
        const int res_mod = p.res_mod > 0 ? p.res_mod : p.M;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wr * 64 + mi * 16 + fr;
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni) {
                const int n = n0 + wc * (BN / 2) + ni * 16 + fq * 4;
                const f32x4 v = finish(p, acc[mi][ni], m, n, res_mod);
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
    static __device__ __forceinline__ f32x4 finish(const Gemm16Args& p, f32x4 v, int m, int n, int res_mod) {
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (p.act == ACT_GELU) {
            v = gelu_erf_fast4(v);      // the same arithmetic in every GEMM kernel: a tile's bits must not depend on which one its batch size selects
        } else if (p.act == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (p.residual) v += *(const f32x4*)(p.residual + (size_t)(m % res_mod) * p.N + n);
        return v;
    }
};

// grouped tile order: G16_GROUP_M row tiles share each W panel back to back
template <int BN>
__device__ __forceinline__ void g2_coords(int t, int tilesM, int tilesN, int& m0, int& n0) {
    const int per_group = G16_GROUP_M * tilesN;
    const int group = t / per_group;
    const int first_m = group * G16_GROUP_M;
    const int gsz = min(G16_GROUP_M, tilesM - first_m);
    const int in_group = t - group * per_group;
    m0 = (first_m + in_group % gsz) * 256;
    n0 = (in_group / gsz) * BN;
}

template <class T, int BN>
__global__ __launch_bounds__(512, 2) void gemm16v2_kernel(Gemm16Args p) {
    using C = G2<BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    G2Core<T, BN> g;
    g.init(smem, p);
    int m0, n0;
    g2_coords<BN>(xcd_remap(blockIdx.x, gridDim.x), p.M / C::BM, p.N / BN, m0, n0);
    const int nk = p.K / C::BK;

    g.stage(0, m0, n0, 0);
    if (nk > 1) g.stage(1, m0, n0, 1);
    int slot = 0, kt = 0;
    for (; kt + 2 < nk; ++kt) {
        g.template ktile<true, 0>(slot, m0, n0, kt + 2, false);
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (kt + 1 < nk) {
        g.template ktile<false, 0>(slot, 0, 0, 0, false);
        slot = slot == 2 ? 0 : slot + 1;
    }
    g.template ktile<false, 2>(slot, 0, 0, 0, false);
    g.epilogue(p, m0, n0);
}

}  // namespace wm
