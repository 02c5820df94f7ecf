// High-frequency component of a tile: MedSAM.fft (segment_anything/network.py:36-57).
//
//   gray = 0.2989 R + 0.587 G + 0.114 B                    (torchvision Grayscale, :41)
//   out  = | Re( ifft2( mask * fft2(gray) ) ) |            (:47-55, norm="forward" both ways)
// where mask zeroes the centred square of signed frequencies [-181, 180]^2
// (line = int((1024*1024*0.125)**.5 // 2) = 181, :44-45).
//
// mask = 1 - lowpass, so out = | gray - Re(ifft2(X restricted to L x L)) | with
// L = 362 frequencies per axis.  Only the L x L block of the spectrum is ever
// formed: rows are transformed and cut to 362 columns, the 362 columns are
// transformed, cut and immediately transformed back, and the final row pass
// expands 362 -> 1024 and subtracts from the recomputed gray value.  Intermediate
// traffic is 35 % of a full complex 2-D FFT.
//
// 1024-point FFT: radix-4 Stockham autosort in LDS, 256 threads = 256 butterflies
// per pass, 5 passes, twiddles from a table computed in double on the host.
#pragma once
#include "wm_common.h"

namespace wm {

constexpr int FFT_N = 1024;
constexpr int FFT_LINE = 181;            // int((N*N*0.125)**0.5 // 2)
constexpr int FFT_L = 2 * FFT_LINE;      // 362 kept signed frequencies: -181 .. 180

// signed frequency of storage column i (0..361) as an unsigned FFT bin
__device__ __forceinline__ int fft_bin(int i) { return (i - FFT_LINE + FFT_N) & (FFT_N - 1); }

// In: sA holds 1024 complex values.  Out: returns the buffer (sA or sB) holding the result.
// SIGN = -1 forward, +1 inverse (unnormalised).  tw[k] = exp(-2 pi i k / 1024).
template <int SIGN>
__device__ __forceinline__ float2* fft1024(float2* sA, float2* sB, const float2* __restrict__ tw, int j) {
    float2* in = sA; float2* out = sB;
#pragma unroll
    for (int Ns = 1; Ns < FFT_N; Ns *= 4) {
        const int kk = j & (Ns - 1);
        const int tstep = kk * (256 / Ns);                     // angle index for r = 1
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 a = in[j + r * 256];
            if (r == 0 || Ns == 1) { v[r] = a; }
            else {
                float2 w = tw[tstep * r];
                if (SIGN > 0) w.y = -w.y;
                v[r] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
            }
        }
        // radix-4 DFT; forward: W = -i, inverse: W = +i
        const float2 s02 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
        const float2 d02 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        const float2 s13 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
        const float2 d13 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        // (-i)*d13 = (d13.y, -d13.x);  (+i)*d13 = (-d13.y, d13.x)
        const float2 rot = SIGN < 0 ? make_float2(d13.y, -d13.x) : make_float2(-d13.y, d13.x);
        const int base = ((j - kk) << 2) + kk;                 // (j / Ns) * Ns * 4 + j % Ns
        out[base] = make_float2(s02.x + s13.x, s02.y + s13.y);
        out[base + Ns] = make_float2(d02.x + rot.x, d02.y + rot.y);
        out[base + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
        out[base + 3 * Ns] = make_float2(d02.x - rot.x, d02.y - rot.y);
        __syncthreads();
        float2* t = in; in = out; out = t;
    }
    return in;
}

__device__ __forceinline__ float gray_of(const float* __restrict__ x, int64_t b, int y, int col) {
#pragma clang fp contract(off)
    const float* p = x + ((b * 3) * FFT_N + y) * (int64_t)FFT_N + col;
    const float r = p[0], g = p[(int64_t)FFT_N * FFT_N], bl = p[2 * (int64_t)FFT_N * FFT_N];
    return (0.2989f * r + 0.587f * g) + 0.114f * bl;
}

// K1: row transforms, keep 362 columns.  grid (1024, B); out R[b][y][362] complex.
__global__ __launch_bounds__(256) void fft_rows_fwd_kernel(const float* __restrict__ x, float2* __restrict__ R,
                                                           const float2* __restrict__ tw) {
    __shared__ float2 sA[FFT_N], sB[FFT_N];
    const int j = threadIdx.x, y = blockIdx.x;
    const int64_t b = blockIdx.y;
#pragma unroll
    for (int r = 0; r < 4; ++r) sA[j + r * 256] = make_float2(gray_of(x, b, y, j + r * 256), 0.f);
    __syncthreads();
    const float2* res = fft1024<-1>(sA, sB, tw, j);
    float2* dst = R + (b * FFT_N + y) * FFT_L;
    for (int i = j; i < FFT_L; i += 256) dst[i] = res[fft_bin(i)];
}

// K2: per kept column: forward along y, cut to 362 rows, inverse along y (in place).
// grid (362, B).
__global__ __launch_bounds__(256) void fft_cols_kernel(float2* __restrict__ R, const float2* __restrict__ tw) {
    __shared__ float2 sA[FFT_N], sB[FFT_N];
    const int j = threadIdx.x, col = blockIdx.x;
    const int64_t b = blockIdx.y;
    float2* base = R + b * FFT_N * FFT_L + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) sA[j + r * 256] = base[(int64_t)(j + r * 256) * FFT_L];
    __syncthreads();
    float2* res = fft1024<-1>(sA, sB, tw, j);
    float2* other = (res == sA) ? sB : sA;
    // keep signed fy in [-181, 180]: bins 0..180 and 843..1023
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int bin = j + r * 256;
        const bool keep = (bin <= FFT_LINE - 1) || (bin >= FFT_N - FFT_LINE);
        if (!keep) res[bin] = make_float2(0.f, 0.f);
    }
    __syncthreads();
    const float2* back = fft1024<1>(res, other, tw, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) base[(int64_t)(j + r * 256) * FFT_L] = back[j + r * 256];
}

// K3: inverse row transform of the 362 kept columns, out = |gray - Re(.)/N^2|.  grid (1024, B).
// x16 / hfc16 (optional): 16-bit NCHW copies of the three input channels and of the result, for the im2col-free patch embeds
// (gemm16_v3.h AMODE 2) -- this pass has both values in registers anyway.
template <class T>
__global__ __launch_bounds__(256) void fft_rows_inv_kernel(const float* __restrict__ x, const float2* __restrict__ R,
                                                           const float2* __restrict__ tw, float* __restrict__ out,
                                                           u16* __restrict__ x16, u16* __restrict__ hfc16) {
    __shared__ float2 sA[FFT_N], sB[FFT_N];
    const int j = threadIdx.x, y = blockIdx.x;
    const int64_t b = blockIdx.y;
#pragma unroll
    for (int r = 0; r < 4; ++r) sA[j + r * 256] = make_float2(0.f, 0.f);
    __syncthreads();
    const float2* src = R + (b * FFT_N + y) * FFT_L;
    for (int i = j; i < FFT_L; i += 256) sA[fft_bin(i)] = src[i];
    __syncthreads();
    const float2* res = fft1024<1>(sA, sB, tw, j);
    const float norm = 1.0f / ((float)FFT_N * (float)FFT_N);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = j + r * 256;
        const float g = gray_of(x, b, y, col);
        const float o = fabsf(g - res[col].x * norm);
        out[(b * FFT_N + y) * (int64_t)FFT_N + col] = o;
        if (x16) {
            const float* px = x + ((b * 3) * FFT_N + y) * (int64_t)FFT_N + col;
            typename T::elem* d = (typename T::elem*)x16 + ((b * 3) * FFT_N + y) * (int64_t)FFT_N + col;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c * (int64_t)FFT_N * FFT_N] = T::from_f32(px[c * (int64_t)FFT_N * FFT_N]);
            ((typename T::elem*)hfc16)[(b * FFT_N + y) * (int64_t)FFT_N + col] = T::from_f32(o);
        }
    }
}

}  // namespace wm
