// Decoder-side fp32 kernels: small multi-head attention and PostProcess + NMS.
#pragma once
#include "wm_common.h"
#include "../../include/wm_hip.h"

namespace wm {

// ---------------------------------------------------------------------------
// softmax(q k^T / sqrt(hd)) v in fp32 (transformer.py:225-238).  One wave per (tile, head, QB queries): lanes
// stride over the keys with a per-lane online softmax per query, merged across the wave at the end.  QB queries
// share every K / V row a lane loads (the token->image attention reads 4096 keys for only 51 queries, so the
// K / V traffic, not the arithmetic, is what costs).  q [B,Nq,heads*HD], k/v [B,Nk,heads*HD].
// With NW > 1 the keys are additionally split over NW waves of the workgroup and merged through LDS.
// grid (ceil(Nq / QB), heads, B), 64 * NW threads.
// ---------------------------------------------------------------------------
template <int HD, int QB, int NW = 1>
__global__ __launch_bounds__(64 * NW) void mha32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, float* __restrict__ out,
                                                   int nq, int nk, int heads) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * QB, head = blockIdx.y, b = blockIdx.z;
    const int C = heads * HD;
    const float scale = 1.0f / sqrtf((float)HD);
    float qv[QB][HD], m[QB], l[QB], acc[QB][HD];
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const int qi = min(q0 + i, nq - 1);                       // tail queries recompute the last one; not stored
        const float* qp = q + ((size_t)b * nq + qi) * C + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = *(const f32x4*)(qp + d);
            qv[i][d] = t[0]; qv[i][d + 1] = t[1]; qv[i][d + 2] = t[2]; qv[i][d + 3] = t[3];
        }
        m[i] = -1e30f; l[i] = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[i][d] = 0.f;
    }
    for (int key = wave * 64 + lane; key < nk; key += 64 * NW) {
        const float* kp = k + ((size_t)b * nk + key) * C + head * HD;
        const float* vp = v + ((size_t)b * nk + key) * C + head * HD;
        float kr[HD], vr[HD];
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = *(const f32x4*)(kp + d);
            const f32x4 u = *(const f32x4*)(vp + d);
            kr[d] = t[0]; kr[d + 1] = t[1]; kr[d + 2] = t[2]; kr[d + 3] = t[3];
            vr[d] = u[0]; vr[d + 1] = u[1]; vr[d + 2] = u[2]; vr[d + 3] = u[3];
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4)
                s += qv[i][d] * kr[d] + qv[i][d + 1] * kr[d + 1] + qv[i][d + 2] * kr[d + 2] + qv[i][d + 3] * kr[d + 3];
            s *= scale;
            const float mn = fmaxf(m[i], s);
            const float a = expf(m[i] - mn), pe = expf(s - mn);
            l[i] = l[i] * a + pe;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[i][d] = acc[i][d] * a + pe * vr[d];
            m[i] = mn;
        }
    }
    __shared__ float part[NW][QB][HD + 2];
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const float M = wave_max(m[i]);
        const float f = expf(m[i] - M);          // lanes that saw no key: m = -1e30 -> f = 0
        const float L = wave_sum(l[i] * f);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const float o = wave_sum(acc[i][d] * f);
            if (lane == 0) part[wave][i][d] = o;
        }
        if (lane == 0) { part[wave][i][HD] = M; part[wave][i][HD + 1] = L; }
    }
    __syncthreads();
    // merge the NW partial softmaxes: thread (i, d) of the first QB*HD threads
    const int t = threadIdx.x;
    if (t < QB * HD) {
        const int i = t / HD, d = t % HD;
        if (q0 + i < nq) {
            float M = part[0][i][HD];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, part[w][i][HD]);
            float L = 0.f, o = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float f = expf(part[w][i][HD] - M);
                L += part[w][i][HD + 1] * f;
                o += part[w][i][d] * f;
            }
            out[((size_t)b * nq + q0 + i) * C + head * HD + d] = o / L;
        }
    }
}

// ---------------------------------------------------------------------------
// Few queries over many keys (token -> image, transformer.py:160-170 and :97-104: 51 queries per tile, 4096 keys), keys split
// over workgroups: mha32_kernel<16, 4, 4> gives every group of 4 queries its own workgroup, so the 512 KB of one (tile, head)'s
// K / V are read 13 times (165 us per launch at 16 tiles, K / V traffic).  Here a workgroup owns KC consecutive keys of one
// (tile, head): it reads them ONCE into LDS; lane = query (nq <= 64), the 4 waves take KC / 4 keys each and read every K / V row
// as an LDS broadcast; the 4 partial softmaxes are merged through LDS and written as (o[HD], m, l) per query and key chunk;
// mha32_merge_chunks_kernel combines the chunks.  grid (nk / KC, heads, B), 256 threads.
// ---------------------------------------------------------------------------
template <int HD, int KC>
__global__ __launch_bounds__(256) void mha32_keysplit_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, float* __restrict__ part,
                                                             int nq, int nk, int heads) {
    static_assert(KC % 4 == 0 && HD % 4 == 0, "geometry");
    __shared__ __attribute__((aligned(16))) float sK[KC * HD];
    __shared__ __attribute__((aligned(16))) float sV[KC * HD];
    __shared__ float sP[4][64][HD + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, head = blockIdx.y, b = blockIdx.z, nchunk = gridDim.x;
    const int C = heads * HD;
    const float scale = 1.44269504088896340736f / sqrtf((float)HD);      // scores in the log2 domain: exp2 is one instruction
    const float* kb = k + ((size_t)b * nk + (size_t)chunk * KC) * C + head * HD;
    const float* vb = v + ((size_t)b * nk + (size_t)chunk * KC) * C + head * HD;
    constexpr int CPR = HD / 4;                               // 16-byte chunks per row
    {   // all of a thread's K / V chunks requested before the first is stored (a rolled loop waits for each load in turn)
        constexpr int NCH = KC * CPR / 256;
        static_assert(KC * CPR % 256 == 0, "chunks per thread");
        f32x4 kx[NCH], vx[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256, row = e / CPR, c4 = e % CPR;
            kx[i] = *(const f32x4*)(kb + (size_t)row * C + c4 * 4);
            vx[i] = *(const f32x4*)(vb + (size_t)row * C + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256, row = e / CPR, c4 = e % CPR;
            *(f32x4*)(sK + row * HD + c4 * 4) = kx[i];
            *(f32x4*)(sV + row * HD + c4 * 4) = vx[i];
        }
    }
    float qv[HD], acc[HD];
    {
        const int qi = min(lane, nq - 1);                     // lanes past nq recompute the last query; not stored
        const float* qp = q + ((size_t)b * nq + qi) * C + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const f32x4 t = *(const f32x4*)(qp + d);
            qv[d] = t[0] * scale; qv[d + 1] = t[1] * scale; qv[d + 2] = t[2] * scale; qv[d + 3] = t[3] * scale;    // scale includes log2 e
        }
    }
    float m = -1e30f, l = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    __syncthreads();
    constexpr int KW = KC / 4, KB = 8;                       // keys per wave; keys per softmax block (one rescale of acc per block)
    static_assert(KW % KB == 0, "key blocks");
    for (int j0 = 0; j0 < KW; j0 += KB) {
        float sc[KB];
        float bm = -1e30f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float* kr = sK + (wave * KW + j0 + j) * HD;    // wave-uniform address: an LDS broadcast
            float t0 = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const f32x4 t = *(const f32x4*)(kr + d);
                t0 = fmaf(qv[d], t[0], t0); t0 = fmaf(qv[d + 1], t[1], t0); t0 = fmaf(qv[d + 2], t[2], t0); t0 = fmaf(qv[d + 3], t[3], t0);
            }
            sc[j] = t0;
            bm = fmaxf(bm, t0);
        }
        const float mn = fmaxf(m, bm);
        const float a = __builtin_amdgcn_exp2f(m - mn);
        l *= a;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] *= a;
        m = mn;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float* vr = sV + (wave * KW + j0 + j) * HD;
            const float pe = __builtin_amdgcn_exp2f(sc[j] - mn);
            l += pe;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const f32x4 u = *(const f32x4*)(vr + d);
                acc[d] = fmaf(pe, u[0], acc[d]); acc[d + 1] = fmaf(pe, u[1], acc[d + 1]);
                acc[d + 2] = fmaf(pe, u[2], acc[d + 2]); acc[d + 3] = fmaf(pe, u[3], acc[d + 3]);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) sP[wave][lane][d] = acc[d];
    sP[wave][lane][HD] = m; sP[wave][lane][HD + 1] = l;
    __syncthreads();
    // merge the 4 waves: thread (query, 4-column group) of the first 64 * HD / 4 threads
    if (tid < 64 * (HD / 4)) {
        const int qi = tid / (HD / 4), d0 = (tid % (HD / 4)) * 4;
        if (qi < nq) {
            float M = sP[0][qi][HD];
#pragma unroll
            for (int w = 1; w < 4; ++w) M = fmaxf(M, sP[w][qi][HD]);
            float L = 0.f;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = __builtin_amdgcn_exp2f(sP[w][qi][HD] - M);
                L += sP[w][qi][HD + 1] * f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) o[jj] += sP[w][qi][d0 + jj] * f;
            }
            float* dst = part + ((((size_t)b * heads + head) * nchunk + chunk) * 64 + qi) * (HD + 2);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) dst[d0 + jj] = o[jj];
            if (d0 == 0) { dst[HD] = M; dst[HD + 1] = L; }
        }
    }
}

// out[b][q][head*HD + d] = sum_chunks o f / sum_chunks l f, f = exp2(m_chunk - max m).  grid (heads, B), 64 * HD / 4 threads
template <int HD>
__global__ void mha32_merge_chunks_kernel(const float* __restrict__ part, float* __restrict__ out, int nq, int nchunk, int heads) {
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int qi = tid / (HD / 4), d0 = (tid % (HD / 4)) * 4;
    if (qi >= nq) return;
    const float* src = part + (((size_t)b * heads + head) * nchunk * 64 + qi) * (HD + 2);
    const size_t cs = (size_t)64 * (HD + 2);
    float M = -1e30f;
    for (int c = 0; c < nchunk; ++c) M = fmaxf(M, src[c * cs + HD]);
    float L = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunk; ++c) {
        const float f = __builtin_amdgcn_exp2f(src[c * cs + HD] - M);      // the partial maxima are log2-domain
        L += src[c * cs + HD + 1] * f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) o[jj] += src[c * cs + d0 + jj] * f;
    }
    const float inv = 1.0f / L;
    float* dst = out + ((size_t)b * nq + qi) * (heads * HD) + head * HD + d0;
    *(f32x4*)dst = f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
}

// ---------------------------------------------------------------------------
// The same attention for many queries over a few keys (image -> token, transformer.py:172-178: 4096 queries per
// tile, NK = 51 keys).  One THREAD per query: K and V rows are uniform across the wave (scalar loads, operands from
// SGPRs), the NK scores stay in registers (two-pass softmax, no rescaling), nothing crosses lanes.
// grid (ceil(Nq / 256), heads, B), 256 threads.
// ---------------------------------------------------------------------------
template <int HD, int NK>
__global__ __launch_bounds__(256) void mha32_fewkeys_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ out,
                                                            int nq, int heads) {
    const int qi = blockIdx.x * 256 + threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    if (qi >= nq) return;
    const int C = heads * HD;
    const float scale = 1.0f / sqrtf((float)HD);
    float qv[HD];
    const float* qp = q + ((size_t)b * nq + qi) * C + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const f32x4 t = *(const f32x4*)(qp + d);
        qv[d] = t[0] * scale; qv[d + 1] = t[1] * scale; qv[d + 2] = t[2] * scale; qv[d + 3] = t[3] * scale;
    }
    const float* kb = k + (size_t)b * NK * C + head * HD;      // wave-uniform
    const float* vb = v + (size_t)b * NK * C + head * HD;
    float sc[NK];
    float m = -1e30f;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(qv[d], kb[(size_t)j * C + d], s);
        sc[j] = s;
        m = fmaxf(m, s);
    }
    float l = 0.f, acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const float pe = expf(sc[j] - m);
        l += pe;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(pe, vb[(size_t)j * C + d], acc[d]);
    }
    const float inv = 1.0f / l;
    float* op = out + ((size_t)b * nq + qi) * C + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) *(f32x4*)(op + d) = f32x4{acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv};
}

// ---------------------------------------------------------------------------
// PostProcess (build_sam.py:219-258) + score cut and greedy NMS
// (visualize_prediction.py:150-157; torchvision.ops.nms semantics: stable
// descending sort, suppress when IoU > thr).  One 64-lane wave per tile, lane =
// query slot.  Arithmetic is kept un-contracted so box/IoU values match a plain
// fp32 CPU evaluation bit for bit.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void postprocess_nms_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ boxes,
                                                             const float* __restrict__ target_sizes,
                                                             float conf_thr, float score_thr, float iou_thr,
                                                             wm_box_record* __restrict__ rec) {
#pragma clang fp contract(off)
    constexpr int NQ = WM_NUM_QUERIES, NL = WM_NUM_LOGITS;
    __shared__ float sx0[64], sy0[64], sx1[64], sy1[64], sarea[64], sscore[64];
    __shared__ int sorder[64], scand[64], sdead[64];
    const int b = blockIdx.x, i = threadIdx.x;
    const bool live = i < NQ;
    float score = 0.f; int label = 0;
    float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f;
    if (live) {
        const float* lg = logits + ((size_t)b * NQ + i) * NL;
        float mx = lg[0];
#pragma unroll
        for (int j = 1; j < NL; ++j) mx = fmaxf(mx, lg[j]);
        float e[NL], sum = 0.f;
#pragma unroll
        for (int j = 0; j < NL; ++j) { e[j] = expf(lg[j] - mx); sum += e[j]; }
        score = e[0] / sum;                         // max over the first NL-1 columns, first index wins ties
#pragma unroll
        for (int j = 1; j < NL - 1; ++j) {
            const float pj = e[j] / sum;
            if (pj > score) { score = pj; label = j; }
        }
        const float* bx = boxes + ((size_t)b * NQ + i) * 4;
        const float cx = bx[0], cy = bx[1], w = bx[2], h = bx[3];
        const float sw = target_sizes[b * 2 + 0], sh = target_sizes[b * 2 + 1];   // build_sam.py:252-253
        x0 = (cx - 0.5f * w) * sw; y0 = (cy - 0.5f * h) * sh;
        x1 = (cx + 0.5f * w) * sw; y1 = (cy + 0.5f * h) * sh;
    }
    const bool conf = live && score > conf_thr;
    const bool cand = conf && score > score_thr;
    sx0[i] = x0; sy0[i] = y0; sx1[i] = x1; sy1[i] = y1;
    sarea[i] = (x1 - x0) * (y1 - y0);
    sscore[i] = score; scand[i] = cand ? 1 : 0; sdead[i] = 0;
    __syncthreads();
    // rank among candidates: descending score, ties by ascending slot (stable sort)
    int rank = 0, ncand = 0;
    for (int j = 0; j < NQ; ++j) {
        if (scand[j]) {
            ++ncand;
            if (sscore[j] > score || (sscore[j] == score && j < i)) ++rank;
        }
    }
    if (cand) sorder[rank] = i;
    __syncthreads();
    int nms_rank = -1, kept = 0;
    for (int r = 0; r < ncand; ++r) {
        const int a = sorder[r];
        const bool alive = sdead[a] == 0;            // uniform across the wave
        if (alive) {
            if (i == a) nms_rank = kept;
            ++kept;
            if (cand && rank > r && sdead[i] == 0) {
                const float xx0 = fmaxf(sx0[a], x0), yy0 = fmaxf(sy0[a], y0);
                const float xx1 = fminf(sx1[a], x1), yy1 = fminf(sy1[a], y1);
                const float iw = fmaxf(0.f, xx1 - xx0), ih = fmaxf(0.f, yy1 - yy0);
                const float inter = iw * ih;
                const float iou = inter / (sarea[a] + sarea[i] - inter);
                if (iou > iou_thr) sdead[i] = 1;
            }
        }
        __syncthreads();
    }
    if (live) {
        wm_box_record o;
        o.box[0] = x0; o.box[1] = y0; o.box[2] = x1; o.box[3] = y1;
        o.score = score; o.label = label;
        o.flags = (conf ? WM_FLAG_CONF : 0) | (cand ? WM_FLAG_SCORE : 0) | (nms_rank >= 0 ? WM_FLAG_NMS : 0);
        o.nms_rank = nms_rank;
        rec[(size_t)b * NQ + i] = o;
    }
}

// ---------------------------------------------------------------------------
// Large-frame front end: merge the detections of overlapping 1024 x 1024 tiles of one frame.
// Input: the per-tile records of postprocess_nms_kernel (boxes in tile pixels) and each tile's origin in the frame.
// Candidates = slots that survived their own tile's NMS; their boxes move to frame coordinates and one more greedy
// class-agnostic NMS (the same arithmetic, un-contracted) runs over all of them: an animal seen by two overlapping tiles
// is reported once.  One workgroup of 1024 threads, up to 4096 slots (80 tiles); candidates are ranked by descending
// score, ties by ascending slot (stable).  Output records: frame-coordinate boxes, WM_FLAG_MERGED set on survivors,
// nms_rank = position in the merged list (-1 otherwise).  No reference behaviour exists for this step (the reference
// down-scales whole frames, dataloader_coco.py:288); the checker is the numpy restatement in oracle/tiling_oracle.py.
// ---------------------------------------------------------------------------
constexpr int MERGE_MAX_SLOTS = 4096, MERGE_THREADS = 1024, MERGE_PER = MERGE_MAX_SLOTS / MERGE_THREADS;

__global__ __launch_bounds__(1024) void merge_tiles_nms_kernel(const wm_box_record* __restrict__ rec, const int* __restrict__ origins,
                                                               int n_slots, float iou_thr, wm_box_record* __restrict__ out) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sx0 = (float*)smem;                 // [MERGE_MAX_SLOTS] each
    float* sy0 = sx0 + MERGE_MAX_SLOTS;
    float* sx1 = sy0 + MERGE_MAX_SLOTS;
    float* sy1 = sx1 + MERGE_MAX_SLOTS;
    float* sscore = sy1 + MERGE_MAX_SLOTS;
    int* sorder = (int*)(sscore + MERGE_MAX_SLOTS);
    unsigned char* scand = (unsigned char*)(sorder + MERGE_MAX_SLOTS);
    unsigned char* sdead = scand + MERGE_MAX_SLOTS;
    int& s_ncand = *(int*)(sdead + MERGE_MAX_SLOTS);           // everything in the dynamic region (16-byte aligned base)
    const int tid = threadIdx.x;
    wm_box_record r[MERGE_PER];
    bool cand[MERGE_PER];
    if (tid == 0) s_ncand = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MERGE_PER; ++k) {
        const int i = tid + k * MERGE_THREADS;
        cand[k] = false;
        if (i < n_slots) {
            r[k] = rec[i];
            const int tile = i / WM_NUM_QUERIES;
            const float oy = (float)origins[2 * tile], ox = (float)origins[2 * tile + 1];
            r[k].box[0] += ox; r[k].box[1] += oy; r[k].box[2] += ox; r[k].box[3] += oy;
            cand[k] = (r[k].flags & WM_FLAG_NMS) != 0;
            sx0[i] = r[k].box[0]; sy0[i] = r[k].box[1]; sx1[i] = r[k].box[2]; sy1[i] = r[k].box[3];
            sscore[i] = r[k].score;
            scand[i] = cand[k] ? 1 : 0;
            sdead[i] = 0;
            if (cand[k]) atomicAdd(&s_ncand, 1);
        }
    }
    __syncthreads();
    const int ncand = s_ncand;
    int rank[MERGE_PER];
#pragma unroll
    for (int k = 0; k < MERGE_PER; ++k) {
        const int i = tid + k * MERGE_THREADS;
        rank[k] = 0;
        if (cand[k]) {
            const float sc = r[k].score;
            int rk = 0;
            for (int j = 0; j < n_slots; ++j)
                if (scand[j] && (sscore[j] > sc || (sscore[j] == sc && j < i))) ++rk;
            rank[k] = rk;
            sorder[rk] = i;
        }
    }
    __syncthreads();
    int merged_rank[MERGE_PER];
#pragma unroll
    for (int k = 0; k < MERGE_PER; ++k) merged_rank[k] = -1;
    int kept = 0;
    for (int rr = 0; rr < ncand; ++rr) {
        const int a = sorder[rr];
        const bool alive = sdead[a] == 0;            // uniform across the workgroup
        if (alive) {
            const float ax0 = sx0[a], ay0 = sy0[a], ax1 = sx1[a], ay1 = sy1[a];
            const float aarea = (ax1 - ax0) * (ay1 - ay0);
#pragma unroll
            for (int k = 0; k < MERGE_PER; ++k) {
                const int i = tid + k * MERGE_THREADS;
                if (i == a) merged_rank[k] = kept;
                if (cand[k] && rank[k] > rr && sdead[i] == 0) {
                    const float xx0 = fmaxf(ax0, r[k].box[0]), yy0 = fmaxf(ay0, r[k].box[1]);
                    const float xx1 = fminf(ax1, r[k].box[2]), yy1 = fminf(ay1, r[k].box[3]);
                    const float iw = fmaxf(0.f, xx1 - xx0), ih = fmaxf(0.f, yy1 - yy0);
                    const float inter = iw * ih;
                    const float area = (r[k].box[2] - r[k].box[0]) * (r[k].box[3] - r[k].box[1]);
                    const float iou = inter / (aarea + area - inter);
                    if (iou > iou_thr) sdead[i] = 1;
                }
            }
            ++kept;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < MERGE_PER; ++k) {
        const int i = tid + k * MERGE_THREADS;
        if (i < n_slots) {
            r[k].flags = (r[k].flags & ~WM_FLAG_MERGED) | (merged_rank[k] >= 0 ? WM_FLAG_MERGED : 0);
            r[k].nms_rank = merged_rank[k];
            out[i] = r[k];
        }
    }
}

// Cut 1024 x 1024 tiles at `origins` (y0, x0) out of one uint8 HWC frame [H,W,3] into the model's input tensor
// [n,3,1024,1024] fp32: ToTensor + Normalize as preprocess_u8_kernel, zeros where a tile reaches past the frame.
__global__ __launch_bounds__(256) void tile_frame_u8_kernel(const unsigned char* __restrict__ frame, const int* __restrict__ origins,
                                                            float* __restrict__ out, int n, int H, int W) {
#pragma clang fp contract(off)
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int64_t total = (int64_t)n * 1024 * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x4 = (int)(i & 255) * 4;
        const int y = (int)((i >> 8) & 1023);
        const int64_t t = i >> 18;
        const int fy = origins[2 * t] + y, fx0 = origins[2 * t + 1] + x4;
        f32x4 v[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (fy >= 0 && fy < H) {
            const unsigned char* row = frame + (int64_t)fy * W * 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int fx = fx0 + j;
                if (fx >= 0 && fx < W) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c][j] = ((float)row[(int64_t)fx * 3 + c] / 255.0f - mean[c]) / stdv[c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) *(f32x4*)(out + ((t * 3 + c) * 1024 + y) * (int64_t)1024 + x4) = v[c];
    }
}

}  // namespace wm
