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
};

struct G8 {
    static constexpr int BM = 256, BN = 256, BKB = 128, NSLOT = 2;
    static constexpr int A_BYTES = BM * BKB, W_BYTES = BN * BKB, STAGE = A_BYTES + W_BYTES;
    static constexpr int LDS = NSLOT * STAGE + 32 * 1024;          // ring + epilogue room (second residual landing buffer)
    static constexpr int MT = 4, NT = 2, KS = BKB / 64;
    static constexpr int P = (BM / 8 + BN / 8) / 8;                // DMA pieces per wave and K-step
};

template <class T>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(Gemm8Args p) {
    using C = G8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
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

    // DMA piece = 8 rows x 128 B; lane -> row (lane >> 3), physical chunk (lane & 7); the row's swizzle key is
    // ((row >> 1) & 7) = (4 (piece & 1) + (lane >> 4)) & 7, so even and odd pieces have their own per-lane source offset
    const size_t row_bytes = (size_t)K;
    unsigned lane_off[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        lane_off[par] = (unsigned)(lane >> 3) * (unsigned)K + (unsigned)((((lane & 7) ^ ((4 * par + (lane >> 4)) & 7))) << 4);
    auto stage = [&](int slot, int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            const char* base = Ab + (size_t)(m0 + piece * 8) * row_bytes + (size_t)s * C::BKB;
            __builtin_amdgcn_global_load_lds(base + lane_off[i & 1], WM_LDS_PTR(smem + slot * C::STAGE + piece * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            const char* base = Wb + (size_t)(n0 + piece * 8) * row_bytes + (size_t)s * C::BKB;
            __builtin_amdgcn_global_load_lds(base + lane_off[i & 1], WM_LDS_PTR(smem + slot * C::STAGE + C::A_BYTES + piece * 1024), 16, 0, 0);
        }
    };

    // fragment read: row r32 of a 32-row tile, chunks 4 ks + 2 h and + 1 -> two ds_read_b128 whose addresses differ by
    // XOR constants only (the swizzle key has no bit in common with them)
    const int frag_off = r32 * C::BKB + ((((2 * h) ^ ((r32 >> 1) & 7))) << 4);
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
    auto mfmas = [&]() {
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
            for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NT; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0, one, 0, one);
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- main loop: two slots, one K-step of DMA in flight.  Both wave groups issue the pieces of step s + 1 right after
    // X_s: slot (s + 1) & 1 was last read for step s - 1, by waves 0-3 before Y_(s-1) and by waves 4-7 between Y_(s-1) and
    // X_s (drained with lgkmcnt(0) before they arrive at X_s).  A wave waits for its own pieces of step s (vmcnt(0): nothing
    // younger is outstanding at that point) before X_s, and every read of slot s & 1 follows X_s.
    stage(0, 0);
    if (wr == 0) {
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            wait_vmcnt<0>();
            barrier();                                      // X_s
            if (s + 1 < ns) stage((s + 1) & 1, s + 1);
            read_frags(s & 1);
            barrier();                                      // Y_s
            mfmas();
        }
    } else {
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of the slot about to be overwritten are back
            wait_vmcnt<0>();
            barrier();                                      // X_s
            if (s + 1 < ns) stage((s + 1) & 1, s + 1);
            if (s > 0) mfmas();                             // step s - 1
            barrier();                                      // Y_s
            read_frags(s & 1);
        }
        mfmas();                                            // step ns - 1
    }

    // ---- epilogue (through LDS so that every global access is row-contiguous; see gemm16_v5.h) ----
    // lane holds, per 32 x 32 tile (mi, ni): row m = r32, columns n = 8 g + 4 h + (0..3) for g = 0..3 (registers 4 g .. 4 g + 3)
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
    barrier();                                              // every wave is done with the ring
    auto scaled = [&](int mi, int ni, int g, const f32x4& sc, const f32x4& bi) {
        f32x4 v{acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], bi[j]);
        return v;
    };
    const int act = p.act & 0xff;
    if (p.residual != nullptr) {
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
                const int rr = piece * 2 + (lane >> 5), ch = lane & 31;
                const int m = m0 + (rr >> 5) * 128 + mi * 32 + (rr & 31);
                __builtin_amdgcn_global_load_lds((const char*)(p.residual + (size_t)m * p.N + col_of(ch, ni)), WM_LDS_PTR(dst + piece * 1024), 16, 0, 0);
            }
        };
        res_dma(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int mi = q >> 1, ni = q & 1;
            if (q + 1 < 8) res_dma(q + 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + 64 * wc + 32 * ni + 8 * g + 4 * h;
                const f32x4 sc = *(const f32x4*)(p.wscale + n);
                const f32x4 bi = p.bias ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4*)(smem + STG + (wr * 32 + r32) * ROWB + (wc * 32 + 8 * g + 4 * h) * 4) = scaled(mi, ni, g, sc, bi);
            }
            if (q + 1 < 8) wait_vmcnt<4>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            barrier();
            const char* land = smem + ((q & 1) ? L1 : L0);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int c = it * 512 + tid, rr = c >> 5, ch = c & 31;
                const int m = m0 + (rr >> 5) * 128 + mi * 32 + (rr & 31);
                const f32x4 v = *(const f32x4*)(smem + STG + rr * ROWB + ch * 16) + *(const f32x4*)(land + c * 16);
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
                const f32x4 sc = *(const f32x4*)(p.wscale + n0 + nl);
                const f32x4 bi = p.bias ? *(const f32x4*)(p.bias + n0 + nl) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < C::MT; ++mi) {
                    f32x4 v = scaled(mi, ni, g, sc, bi);
                    if (act == ACT_GELU) v = gelu_erf_fast4(v);
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
        // 16-bit output (qkv): 2 passes of 128 rows (mi = 2 q, 2 q + 1 of both wave rows) x 256 columns
        constexpr int ROWB = 256 * 2 + 16;
        static_assert(128 * ROWB <= C::LDS, "epilogue LDS map");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int ni = 0; ni < C::NT; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = 64 * wc + 32 * ni + 8 * g + 4 * h;
                    const f32x4 sc = *(const f32x4*)(p.wscale + n0 + nl);
                    const f32x4 bi = p.bias ? *(const f32x4*)(p.bias + n0 + nl) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        f32x4 v = scaled(2 * q + mm, ni, g, sc, bi);
                        if (act == ACT_GELU) v = gelu_erf_fast4(v);
                        else if (act == ACT_RELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                        typename T::vec4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                        *(typename T::vec4*)(smem + (wr * 64 + mm * 32 + r32) * ROWB + nl * 2) = o;
                    }
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = it * 512 + tid, rr = c >> 5, ch = c & 31;
                const int m = m0 + (rr >> 6) * 128 + q * 64 + (rr & 63);
                const f32x4 v = *(const f32x4*)(smem + rr * ROWB + ch * 16);
                *(f32x4*)((char*)p.out16 + ((size_t)m * p.N + n0) * 2 + ch * 16) = v;
            }
            if (q == 0) __syncthreads();
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
