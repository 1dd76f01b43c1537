// FP64 MFMA issue-rate / clock probe for gfx950.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(int iters, unsigned long long* cyc, unsigned long long* rt, double* sink) {
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        cyc[w] = t1 - t0;
        rt[w] = r1 - r0;
    }
}

template <int NACC>
void run(int blocks, int threads, int iters) {
    int waves = blocks * threads / 64;
    unsigned long long *cyc, *rt; double* sink;
    hipMalloc(&cyc, waves * 8); hipMalloc(&rt, waves * 8); hipMalloc(&sink, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, 100, cyc, rt, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, iters, cyc, rt, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> hc(waves), hr(waves);
    hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), rt, waves * 8, hipMemcpyDeviceToHost);
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    double mc = hc[waves / 2], mr = hr[waves / 2];
    double nm = (double)iters * NACC;
    double flops = (double)waves * nm * 2048.0;
    printf("NACC=%d blocks=%d thr=%d: %.3f ms  %.1f TFLOP/s | per wave: %.1f cyc/MFMA, clock %.3f GHz (memtime/realtime@100MHz)\n",
           NACC, blocks, threads, ms, flops / ms / 1e9, mc / nm, mc / mr * 0.1);
    hipFree(cyc); hipFree(rt); hipFree(sink);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, cus, p.clockRate);
    const int it = 20000;
    run<1>(cus, 256, it);      // 1 wave / SIMD, dependent chain
    run<2>(cus, 256, it);
    run<4>(cus, 256, it);
    run<8>(cus, 256, it);
    run<8>(cus * 2, 256, it);  // 2 waves / SIMD
    run<4>(cus * 2, 256, it);
    run<4>(cus * 3, 256, it);  // 3 waves / SIMD
    run<4>(cus * 4, 256, it);  // 4 waves / SIMD
    run<2>(cus * 4, 256, it);
    run<1>(cus * 4, 256, it);
    run<2>(cus * 8, 256, it);  // 8 waves / SIMD
    run<1>(cus * 8, 256, it);
    run<4>(cus * 2, 512, it);  // 512-thread blocks, 4 waves / SIMD
    return 0;
}
