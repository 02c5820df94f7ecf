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
// per SIMD", item 9).  Ring, swizzle, counted vmcnt, tile order and epilogue are gemm16_v3.h's:
//   RAW  a wave waits for its own pieces of step s (vmcnt) before X_s; every read of slot s follows X_s.
//   WAR  slot (s+2)%3 = (s-1)%3 is overwritten after X_s: waves 0-3 read it before Y_s-1, waves 4-7 after
//        Y_s-1 and drain those reads (lgkmcnt(0)) before they arrive at X_s.
// Needs K / 32 >= 2.  No implicit-conv A mode (gemm16_v3.h keeps that).
#pragma once
#include "gemm16_v3.h"

namespace wm {

template <class T, int BN>
__global__ __launch_bounds__(512, 2) void gemm16v5_kernel(Gemm16Args p) {
    using C = G3<BN, 4>;
    static_assert(C::W_REM == 0 || C::W_REM == 4, "remainder pieces must fall on one wave group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;               // wave group = wr: rows 0-127 / 128-255
    const int fr = lane & 15, fq = lane >> 4;
    const int K = p.K, ns = K / C::BK;
    const char* Ab = (const char*)p.A;
    const char* Wb = (const char*)p.W;

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

    const unsigned lane_off = (unsigned)(lane >> 2) * (unsigned)(K * 2) + (unsigned)((((lane & 3) ^ ((0 - (lane >> 4)) & 3))) << 4);
    const size_t row_bytes = (size_t)K * 2;
    // DMA pieces of this wave for K-step s into ring slot `slot`; EXTRA: waves 0-3 also carry the remainder W piece
    auto stage = [&](int slot, int s, auto extra_tag) {
        constexpr bool EXTRA = decltype(extra_tag)::value;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) {
            const int seg = wave * C::A_PIECES + i;
            const char* base = Ab + (size_t)(m0 + seg * 16) * row_bytes + (size_t)s * 64;
            __builtin_amdgcn_global_load_lds(base + lane_off, WM_LDS_PTR(smem + slot * C::STAGE + seg * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::W_LO + (EXTRA ? 1 : 0); ++i) {
            const int seg = i < C::W_LO ? wave * C::W_LO + i : C::WAVES * C::W_LO + wave;
            const char* base = Wb + (size_t)(n0 + seg * 16) * row_bytes + (size_t)s * 64;
            __builtin_amdgcn_global_load_lds(base + lane_off, WM_LDS_PTR(smem + slot * C::STAGE + C::A_BYTES + seg * 1024), 16, 0, 0);
        }
    };

    const int frag_off = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);
    const int rd_a = (wr * 128) * 64 + frag_off;
    const int rd_w = C::A_BYTES + (wc * C::WCOLS) * 64 + frag_off;

    f32x4 acc[C::MT][C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    typename T::vec8 wf[C::NT], af[C::MT];

    auto read_frags = [&](int slot) {
        const char* sS = smem + slot * C::STAGE;
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
    auto inc = [](int v) { return v == 2 ? 0 : v + 1; };
    auto wait_step = [&](int s, auto extra_tag) {          // this wave's pieces of step s have landed
        constexpr int P = C::P_LO + (decltype(extra_tag)::value ? 1 : 0);
        if (s + 1 < ns) wait_vmcnt<P>(); else wait_vmcnt<0>();
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (wr == 0) {
        using EX = std::integral_constant<bool, (C::W_REM > 0)>;
        stage(0, 0, EX{});
        if (ns > 1) stage(1, 1, EX{});
        int slot = 0;
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            wait_step(s, EX{});
            barrier();                                      // X_s
            read_frags(slot);
            if (s + 2 < ns) stage(slot == 0 ? 2 : slot - 1, s + 2, EX{});
            barrier();                                      // Y_s
            mfmas();
            slot = inc(slot);
        }
    } else {
        using EX = std::false_type;
        stage(0, 0, EX{});
        if (ns > 1) stage(1, 1, EX{});
        int slot = 0;
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of the slot about to be overwritten are back
            wait_step(s, EX{});
            barrier();                                      // X_s
            if (s > 0) mfmas();                             // step s-1
            barrier();                                      // Y_s
            read_frags(slot);
            if (s + 2 < ns) stage(slot == 0 ? 2 : slot - 1, s + 2, EX{});
            slot = inc(slot);
        }
        mfmas();                                            // step ns-1
    }

    const int res_mod = p.res_mod > 0 ? p.res_mod : p.M;
    const int act = p.act & 0xff;
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi) {
        const int m = m0 + wr * 128 + mi * 16 + fr;
#pragma unroll
        for (int ni = 0; ni < C::NT; ++ni) {
            const int n = n0 + wc * C::WCOLS + ni * 16 + fq * 4;
            f32x4 v = acc[mi][ni];
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf_fast(v[j]);
            } else if (act == ACT_RELU) {
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
