// Probe (dev tool): operand lane maps of the block-scaled fp8 MFMAs and the f32 -> e4m3 convert on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/fp8_probe.hip -o tools/fp8_probe && tools/fp8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// e4m3fn decode on the host
static float e4m3_to_f32(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

// A: [32][64] bytes (row-major, k contiguous), B given as Bt: [32 cols][64 k] bytes.  Hypothesis: lane l holds row/col l & 31, k = 32 (l >> 5) + j.
__global__ void k32(const unsigned char* A, const unsigned char* Bt, float* D, unsigned sa, unsigned sb) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a = *(const i32x8*)(A + r * 64 + 32 * h), b = *(const i32x8*)(Bt + r * 64 + 32 * h);
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];      // row = A row, col = lane & 31
}
// 16x16x128: lane l holds row/col l & 15, k = 32 (l >> 4) + j
__global__ void k16(const unsigned char* A, const unsigned char* Bt, float* D, unsigned sa, unsigned sb) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    i32x8 a = *(const i32x8*)(A + r * 128 + 32 * q), b = *(const i32x8*)(Bt + r * 128 + 32 * q);
    f32x4 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 4; ++i) D[(4 * q + i) * 16 + r] = acc[i];
}
__global__ void cvt(const float* x, unsigned char* o, int n) {
    const int i = threadIdx.x;
    if (i * 2 + 1 < n) {
        unsigned r = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0u, false);
        o[2 * i] = r & 0xff;
        o[2 * i + 1] = (r >> 8) & 0xff;
    }
}

int main() {
    srand(1);
    // ---- 32x32x64 ----
    {
        std::vector<unsigned char> A(32 * 64), Bt(32 * 64);
        for (auto& v : A) v = (unsigned char)(rand() % 256);
        for (auto& v : Bt) v = (unsigned char)(rand() % 256);
        for (auto& v : A) if ((v & 0x7f) == 0x7f) v = 0x38;      // no NaN
        for (auto& v : Bt) if ((v & 0x7f) == 0x7f) v = 0x38;
        for (auto& v : A) v = (v & 0x87) | 0x30 | (v & 0x08);    // exponents 6..7: values 0.5 .. 1.9, sums exact enough in fp32
        for (auto& v : Bt) v = (v & 0x87) | 0x30 | (v & 0x08);
        unsigned char *dA, *dB; float* dD;
        hipMalloc(&dA, A.size()); hipMalloc(&dB, Bt.size()); hipMalloc(&dD, 32 * 32 * 4);
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice);
        for (unsigned sc : {0x7f7f7f7fu, 0x80808080u, 0x7f7f7f80u}) {
            hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD, sc, 0x7f7f7f7fu);
            std::vector<float> D(32 * 32);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double err = 0, ref0 = 0;
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int k = 0; k < 64; ++k) s += (double)e4m3_to_f32(A[i * 64 + k]) * e4m3_to_f32(Bt[j * 64 + k]);
                err = fmax(err, fabs(D[i * 32 + j] - s));
                if (i == 0 && j == 0) ref0 = s;
            }
            printf("32x32x64  scaleA=%08x: max |D - ref(scale 1)| = %g   D[0][0]=%g ref=%g ratio=%g\n", sc, err, D[0], ref0, D[0] / ref0);
        }
    }
    // ---- 16x16x128 ----
    {
        std::vector<unsigned char> A(16 * 128), Bt(16 * 128);
        for (auto& v : A) { v = (unsigned char)(rand() % 256); v = (v & 0x8f) | 0x30; }
        for (auto& v : Bt) { v = (unsigned char)(rand() % 256); v = (v & 0x8f) | 0x30; }
        unsigned char *dA, *dB; float* dD;
        hipMalloc(&dA, A.size()); hipMalloc(&dB, Bt.size()); hipMalloc(&dD, 16 * 16 * 4);
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0x7f7f7f7fu, 0x7f7f7f7fu);
        std::vector<float> D(16 * 16);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 128; ++k) s += (double)e4m3_to_f32(A[i * 128 + k]) * e4m3_to_f32(Bt[j * 128 + k]);
            err = fmax(err, fabs(D[i * 16 + j] - s));
        }
        printf("16x16x128 scale 1: max |D - ref| = %g\n", err);
    }
    // ---- convert ----
    {
        const float xs[] = {0.1f, 0.3f, 1.0f, 1.0625f, 1.1875f, 447.f, 448.f, 449.f, 463.9f, 464.f, 465.f, 500.f, 1e6f, -1e6f, 0.001953125f, 0.0009765625f, 0.0029296875f, -0.0f, INFINITY, NAN};
        const int n = sizeof(xs) / 4;
        float* dx; unsigned char* dout;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, n);
        hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, dout, n);
        unsigned char o[64];
        hipMemcpy(o, dout, n, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) printf("cvt %-14g -> 0x%02x = %g\n", xs[i], o[i], e4m3_to_f32(o[i]));
    }
    return 0;
}
