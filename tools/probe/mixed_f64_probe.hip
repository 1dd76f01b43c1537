// Do the FP64 vector pipe and the FP64 matrix pipe of a gfx950 SIMD run side by side?  Workgroups of 8 waves (2 per SIMD): `mode` 0 = every wave
// issues v_fma_f64 chains, 1 = every wave issues v_mfma_f64_16x16x4 chains, 2 = waves alternate (each SIMD holds one of each).  Prints the FP64
// rate of each kind and their sum.   hipcc -O3 --offload-arch=gfx950 -o /tmp/mixed tools/probe/mixed_f64_probe.hip && /tmp/mixed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void probe(int iters, int mode, double* sink) {
    const int wave = threadIdx.x >> 6;
    const bool mfma = mode == 1 || (mode == 2 && ((wave >> 2) & 1));   // waves 0-3 sit on SIMDs 0-3, waves 4-7 again: one of each kind per SIMD
    double s = 0.0;
    if (mfma) {
        v4d acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (v4d){1.0, 2.0, 3.0, 4.0};
        const double a = 1.0 + 1e-12 * threadIdx.x, b = 1e-13 * threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = 1.0 + i + 1e-9 * threadIdx.x;
        const double a = 1.0 + 1e-12 * threadIdx.x, b = 1e-13 * threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)   // 16 FMAs = 2048 flop per wave and trip, as much as one MFMA
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        for (int i = 0; i < 8; ++i) s += acc[i];
    }
    if (s == 12345.678) sink[0] = s;
}
static double run(int mode, int iters, int blocks) {
    double* sink; (void)hipMalloc(&sink, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, 100, mode, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, iters, mode, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(sink);
    return ms;
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount, iters = 20000;
    // per trip: a VALU wave 16 FMAs x 64 lanes x 2 = 2048 flop; an MFMA wave 4 x 2048 flop
    const double fv = (double)blocks * 8 * iters * 2048.0, fm = (double)blocks * 8 * iters * 4 * 2048.0;
    const double t0 = run(0, iters, blocks), t1 = run(1, iters, blocks), t2 = run(2, iters, blocks);
    printf("vector only : %.3f ms  %.1f TFLOP/s\n", t0, fv / t0 / 1e9);
    printf("matrix only : %.3f ms  %.1f TFLOP/s\n", t1, fm / t1 / 1e9);
    printf("half / half : %.3f ms  vector %.1f + matrix %.1f = %.1f TFLOP/s  (alone, with half the waves each: %.3f and %.3f ms)\n", t2, fv / 2 / t2 / 1e9, fm / 2 / t2 / 1e9,
           (fv + fm) / 2 / t2 / 1e9, t0 / 2, t1 / 2);
    return 0;
}
