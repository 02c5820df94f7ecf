// Dev microbenchmark: back-to-back kernel launch cost on one stream (what a kernel pays outside its own work).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ __launch_bounds__(512) void k_lds(float* out, int store) {
    extern __shared__ char smem[];
    if (store) ((float4*)out)[(size_t)blockIdx.x * 512 + threadIdx.x] = float4{1, 2, 3, 4};
    if (threadIdx.x == 1000) out[0] = smem[0];
}
__global__ __launch_bounds__(512) void k_spin(float* out, long long ns) {    // every workgroup busy for `ns`
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ns / 10) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 1000) out[0] = 1;
}
int main() {
    float* out; hipMalloc(&out, 1 << 28);
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 151552);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* what, auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 500; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-70s %7.2f us per launch\n", what, ms * 1e3 / 500);
    };
    timeit("empty <<<1,64>>>", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); });
    timeit("empty <<<256,512>>>, 148 KB LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 151552, 0, out, 0); });
    timeit("empty <<<768,512>>>, 148 KB LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(768), dim3(512), 151552, 0, out, 0); });
    timeit("<<<768,512>>>, 148 KB LDS, 16 B store per thread (6 MB)", [&] { hipLaunchKernelGGL(k_lds, dim3(768), dim3(512), 151552, 0, out, 1); });
    timeit("<<<8192,512>>>, 16 B store per thread (64 MB)", [&] { hipLaunchKernelGGL(k_lds, dim3(8192), dim3(512), 0, 0, out, 1); });
    timeit("spin 50 us <<<256,512>>>", [&] { hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 0, 0, out, 50000LL); });
    timeit("spin 50 us <<<768,512>>> 148 KB LDS (3 rounds)", [&] { hipLaunchKernelGGL(k_spin, dim3(768), dim3(512), 151552, 0, out, 50000LL); });
    return 0;
}
