// v_fma_f64 issue rate on gfx950: N independent chains per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int NCH>
__global__ __launch_bounds__(1024) void probe(int iters, unsigned long long* cyc, double* sink) {
    double acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = 1.0 + i + 1e-9 * threadIdx.x;
    double a = 1.0 + 1e-12 * threadIdx.x, b = 1e-13 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += acc[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NCH>
void run(int blocks, int threads, int iters) {
    int waves = blocks * threads / 64;
    unsigned long long* cyc; double* sink;
    (void)hipMalloc(&cyc, waves * 8); (void)hipMalloc(&sink, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(threads), 0, 0, 100, cyc, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(threads), 0, 0, iters, cyc, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> hc(waves);
    (void)hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    std::sort(hc.begin(), hc.end());
    double n = (double)iters * NCH;
    printf("chains=%d waves/SIMD=%.1f: %.3f ms, %.2f cyc/DFMA/wave, %.1f TFLOP/s\n", NCH, threads / 256.0 * blocks / 256.0, ms,
           hc[waves / 2] / n, waves * n * 128 / ms / 1e9);
}
int main() {
    const int it = 20000;
    run<1>(256, 256, it); run<2>(256, 256, it); run<4>(256, 256, it); run<8>(256, 256, it); run<16>(256, 256, it);
    run<8>(256, 512, it); run<8>(256, 1024, it); run<2>(256, 1024, it); run<1>(256, 1024, it); run<4>(512, 1024, it);
    return 0;
}
