// Global attention, second generation (round 2): one wave per SIMD with the whole 512-entry register file, each wave owning
// TWO 32-query blocks (64 consecutive queries = one grid row when REL).
//
// Why (MI355X_MICROARCH.md, "vector-instruction ISSUE cost" / "one wave per SIMD" rows; tools/overlap.hip): vector work hides
// behind matrix work only inside ONE wave's instruction stream (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles;
// two co-resident waves' MFMA and VALU times add up).  attn_global_kernel (attn16.h) runs two waves per SIMD whose QK^T,
// softmax and P V phases are serial per wave: matrix pipe 36 % + vector ALU 63 % of the time.  Here the two query blocks of
// a wave give the scheduler two independent chains in one stream (block B's exponentials between block A's MFMAs), every
// K and V^T fragment read from LDS feeds two MFMAs instead of one, and a workgroup stages each K/V tile once for 256
// queries instead of 128.
//
// Layout, MFMA maps, rel-pos handling, deferred-max online softmax and the LSUM_IN_O denominator are those of attn16.h.
// grid = (nq / 256) * heads * batch workgroups (1-D, XCD-remapped so that the 16 query blocks of one (tile, head) run on
// one XCD and share its L2 copy of that head's K and V), 256 threads.
#pragma once
#include "attn16.h"

namespace wm {

template <int HD, bool REL> struct Global2Lds {
    using G = AttnGeom<HD>;
    static constexpr int WAVE_F = 64 * 64;                                    // floats per wave: rel_h term [kh][64 queries]; aliased: [32 queries][65] staging
    static constexpr int RELH_BYTES = REL ? 4 * WAVE_F * 4 : 0;
    static constexpr int K_BYTES = 64 * G::KS, V_BYTES = 64 * G::VS;
    static constexpr int KV_OFF = RELH_BYTES;
    static constexpr int TOTAL = RELH_BYTES + 2 * (K_BYTES + V_BYTES);
    static_assert(!REL || 128 * G::KS <= 2 * (K_BYTES + V_BYTES), "table image must fit the K/V ring");
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

// S^T for two query blocks: every K fragment read feeds both
template <class T, int HD>
__device__ __forceinline__ void qk_tile2(f32x16 (&sa)[2], f32x16 (&sb)[2], const typename T::vec8 (&qa)[AttnGeom<HD>::NKS],
                                         const typename T::vec8 (&qb)[AttnGeom<HD>::NKS], const char* sK, int lane) {
    using G = AttnGeom<HD>;
    const int r31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            typename T::vec8 kf = lds_read_v8<T>(sK + (32 * t + r31) * G::KS + (16 * ks + 8 * h) * 2);
            sa[t] = T::mfma32(kf, qa[ks], sa[t]);
            sb[t] = T::mfma32(kf, qb[ks], sb[t]);
        }
}

// running-max bookkeeping of one block for one full 64-key tile (as softmax_pv, attn16.h); returns the exp2 offset
template <int NDT>
__device__ __forceinline__ float softmax_ref_point(SoftmaxState<NDT>& st, const f32x16 (&s)[2], float c1, float tile_bias) {
    float mx0 = -1e30f, mx1 = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        mx0 = fmaxf(mx0, s[0][r]);
        mx1 = fmaxf(mx1, s[1][r]);
    }
    float mx = (fmaxf(mx0, mx1) + tile_bias) * c1;
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float m_use = st.m;
    if (!__all(mx - st.m <= RESCALE_THR)) {
        const float m_new = fmaxf(st.m, mx);
        const float alpha = __builtin_amdgcn_exp2f(st.m - m_new);
        st.l *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[dt][r] *= alpha;
        st.m = m_new;
        m_use = m_new;
    }
    return tile_bias * c1 - m_use;
}

template <class T, int HD, bool REL>
__global__ __launch_bounds__(256, 1) void attn_global2_kernel(AttnArgs p, int nqb) {
    using G = AttnGeom<HD>;
    using L = Global2Lds<HD, REL>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    // logical workgroup id -> (query block, head, tile)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = wg % nqb, head = (wg / nqb) % p.heads, b = wg / (nqb * p.heads);
    const int q0 = qblk * 256 + wave * 64;                    // this wave: queries q0 .. q0 + 63 (block A: +0..31, block B: +32..63)
    const float c1 = p.scale * 1.44269504088896340736f;

    const u16* qb = p.q + ((size_t)b * p.nq) * p.q_stride + head * HD;
    const u16* kb = p.k + ((size_t)b * p.nk) * p.k_stride + head * HD;
    const u16* vb = p.v + ((size_t)b * p.nk) * p.v_stride + head * HD;

    typename T::vec8 qfa[G::NKS], qfb[G::NKS];
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) {
        qfa[ks] = *(const typename T::vec8*)(qb + (size_t)(q0 + c) * p.q_stride + 16 * ks + 8 * h);
        qfb[ks] = *(const typename T::vec8*)(qb + (size_t)(q0 + 32 + c) * p.q_stride + 16 * ks + 8 * h);
    }

    char* sKV = smem + L::KV_OFF;
    f32x16 relwa[2], relwb[2];
    float* sRelH = (float*)smem + wave * L::WAVE_F;           // [kh][64 queries of this wave]

    if constexpr (REL) {
        // ---- prologue: the kw-term (registers) and the kh-term (LDS) of both blocks; table image: 128 rows x HD 16-bit, stride KS
        const int qh = q0 >> 6;                               // q0 is a multiple of 64: both blocks lie in grid row qh
        const float inv_scale = 1.0f / p.scale;
        char* sTab = sKV;
        float* sT = (float*)smem + wave * L::WAVE_F;          // [query c][65] fp32 staging
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const float* tab = which == 0 ? p.rel_w : p.rel_h;
            __syncthreads();
            for (int e = tid; e < 128 * (HD / 4); e += 256) {
                const int row = e / (HD / 4), c4 = e % (HD / 4);
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row < 127) v = *(const f32x4*)(tab + (size_t)row * HD + c4 * 4);
                typename T::vec4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = T::from_f32(v[j]);
                *(typename T::vec4*)(sTab + row * G::KS + c4 * 8) = o;
            }
            __syncthreads();
            if (which == 0) {
                // T_w^T[i][c] = Rw[i].q_c in two passes of 64 table rows; this lane's values are i = qw + 63 - kw
                auto relw_block = [&](f32x16 (&relw)[2], const typename T::vec8 (&qf)[G::NKS], int qw) {
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
                        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS RAW
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int kw = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                                const int idx = qw + 63 - kw;
                                if ((idx >> 6) == pass) relw[t][r] = sT[c * 65 + (idx & 63)] * inv_scale;
                            }
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                    }
                };
                relw_block(relwa, qfa, c);
                relw_block(relwb, qfb, 32 + c);
            } else {
                // T_h^T[kh][c] = Rh[qh + 63 - kh].q_c : table rows taken in reversed order, each fragment used for both blocks
                f32x16 acca[2], accb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acca[t][r] = 0.f; accb[t][r] = 0.f; }
                const int r31 = lane & 31;
#pragma unroll
                for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = qh + 63 - (32 * t + r31);
                        typename T::vec8 kf = lds_read_v8<T>(sTab + row * G::KS + (16 * ks + 8 * h) * 2);
                        acca[t] = T::mfma32(kf, qfa[ks], acca[t]);
                        accb[t] = T::mfma32(kf, qfb[ks], accb[t]);
                    }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kh = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                        sRelH[kh * 64 + c] = acca[t][r] * inv_scale;
                        sRelH[kh * 64 + 32 + c] = accb[t][r] * inv_scale;
                    }
            }
        }
        __syncthreads();
    }

    // ---- main loop over key tiles of 64 (register-staged double buffer, as attn_global_kernel) ----
    const int ntiles = p.nk / 64;
    constexpr int NCH = 64 * G::CH;
    constexpr int PER = (NCH + 255) / 256;
    s16x8 kreg[PER], vreg[PER];
    const u16* kp[PER];
    const u16* vp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = min(tid + i * 256, NCH - 1);
        kp[i] = kb + (size_t)(e / G::CH) * p.k_stride + (e % G::CH) * 8;
        vp[i] = vb + (size_t)(e / G::CH) * p.v_stride + (e % G::CH) * 8;
    }
    const size_t k_step = (size_t)64 * p.k_stride, v_step = (size_t)64 * p.v_stride;
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (tid + i * 256 < NCH) {
                kreg[i] = *(const s16x8*)kp[i];
                vreg[i] = *(const s16x8*)vp[i];
            }
            kp[i] += k_step;
            vp[i] += v_step;
        }
    };
    v_pad_ones<T, HD>(sKV + L::K_BYTES, 64, tid, 256);
    v_pad_ones<T, HD>(sKV + (L::K_BYTES + L::V_BYTES) + L::K_BYTES, 64, tid, 256);
    auto commit = [&](int buf) {
        char* sK = sKV + buf * (L::K_BYTES + L::V_BYTES);
        char* sV = sK + L::K_BYTES;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * 256;
            if (e < NCH) {
                const int key = e / G::CH, ch = e % G::CH;
                *(s16x8*)(sK + key * G::KS + ch * 16) = kreg[i];
                *(s16x8*)(sV + key * G::VS + ch * 16) = vreg[i];
            }
        }
    };

    SoftmaxState<G::NDT> sta, stb;
    sta.init();
    stb.init();
    issue();
    commit(0);
    __syncthreads();

    const int g = lane >> 4;
    const int lq = (lane & 15) >> 2, lp = lane & 3;
    const int v_lane_off = (4 * (g >> 1) + lq) * G::VS + (16 * (g & 1) + 4 * lp) * 2;

    for (int j = 0; j < ntiles; ++j) {
        const int buf = j & 1;
        if (j + 1 < ntiles) issue();
        const char* sK = sKV + buf * (L::K_BYTES + L::V_BYTES);
        const char* sV = sK + L::K_BYTES;
        f32x16 sa[2], sb[2];
        float rha = 0.f, rhb = 0.f;
        if constexpr (REL) {
            rha = sRelH[j * 64 + c];
            rhb = sRelH[j * 64 + 32 + c];
#pragma unroll
            for (int t = 0; t < 2; ++t) { sa[t] = relwa[t]; sb[t] = relwb[t]; }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { sa[t][r] = 0.f; sb[t][r] = 0.f; }
        }
        qk_tile2<T, HD>(sa, sb, qfa, qfb, sK, lane);
        const float offa = softmax_ref_point<G::NDT>(sta, sa, c1, rha);
        const float offb = softmax_ref_point<G::NDT>(stb, sb, c1, rhb);
        float lsa = 0.f, lsb = 0.f;
        // P^T fragments of one 16-key step and O^T += V^T P^T for both blocks (each V^T fragment read feeds two MFMAs)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            typename T::vec8 pa, pb;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float va = __builtin_amdgcn_exp2f(fmaf(sa[ks >> 1][8 * (ks & 1) + jj], c1, offa));
                const float vb2 = __builtin_amdgcn_exp2f(fmaf(sb[ks >> 1][8 * (ks & 1) + jj], c1, offb));
                pa[jj] = T::from_f32(va);
                pb[jj] = T::from_f32(vb2);
                if constexpr (!G::LSUM_IN_O) { lsa += va; lsb += vb2; }
            }
#pragma unroll
            for (int dt = 0; dt < G::NDT; ++dt) {
                const char* pv = sV + (16 * ks) * G::VS + dt * 64 + v_lane_off;
                typename T::vec8 vf = lds_read_vT<T>(pv, 8 * G::VS);
                sta.o[dt] = T::mfma32(vf, pa, sta.o[dt]);
                stb.o[dt] = T::mfma32(vf, pb, stb.o[dt]);
            }
        }
        sta.l += lsa;
        stb.l += lsb;
        if (j + 1 < ntiles) commit(buf ^ 1);
        __syncthreads();
    }
    const size_t rowa = (size_t)b * p.nq + q0 + c, rowb = rowa + 32;
    store_out<T, HD>(sta, p.out + rowa * p.out_stride + head * HD, lane, true, p.out8 ? p.out8 + rowa * p.out_stride + head * HD : nullptr);
    store_out<T, HD>(stb, p.out + rowb * p.out_stride + head * HD, lane, true, p.out8 ? p.out8 + rowb * p.out_stride + head * HD : nullptr);
}

}  // namespace wm
