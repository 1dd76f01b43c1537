// Layout + rate probe for v_mfma_f64_4x4x4_4b_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void layout(const double* A, const double* B, double* D) {
    int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
    D[l] = d;
}
__global__ __launch_bounds__(256) void rate(int iters, unsigned long long* cyc, double* sink) {
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 1234.5) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
int main() {
    // layout: A[l] = unique value per lane; find which (lane_a, lane_b) products feed each output lane
    std::vector<double> hA(64), hB(64), hD(64);
    double *dA, *dB, *dD; (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 512);
    // test 1: A = 1 everywhere, B[l] = 2^(l%16)... decode with one-hot probes instead
    // For each output lane o, determine contributing A lanes: set A = onehot(la), B = all ones -> D[o] != 0
    printf("A-lane -> output lanes (B = 1):\n");
    for (int la = 0; la < 64; ++la) {
        for (int i = 0; i < 64; ++i) { hA[i] = (i == la); hB[i] = 1.0; }
        (void)hipMemcpy(dA, hA.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD.data(), dD, 512, hipMemcpyDeviceToHost);
        printf("A%2d:", la); for (int o = 0; o < 64; ++o) if (hD[o] != 0) printf(" %d", o); printf("\n");
    }
    printf("B-lane -> output lanes (A = 1):\n");
    for (int lb = 0; lb < 64; ++lb) {
        for (int i = 0; i < 64; ++i) { hB[i] = (i == lb); hA[i] = 1.0; }
        (void)hipMemcpy(dA, hA.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD.data(), dD, 512, hipMemcpyDeviceToHost);
        printf("B%2d:", lb); for (int o = 0; o < 64; ++o) if (hD[o] != 0) printf(" %d", o); printf("\n");
    }
    // which (A lane, B lane) pairs are multiplied together: A onehot la, B onehot lb -> any output
    printf("pairs multiplied (la,lb)->o for la<8:\n");
    for (int la = 0; la < 8; ++la) for (int lb = 0; lb < 64; ++lb) {
        for (int i = 0; i < 64; ++i) { hA[i] = (i == la); hB[i] = (i == lb); }
        (void)hipMemcpy(dA, hA.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD.data(), dD, 512, hipMemcpyDeviceToHost);
        for (int o = 0; o < 64; ++o) if (hD[o] != 0) printf("(%d,%d)->%d ", la, lb, o);
    }
    printf("\n");
    // rate
    int blocks = 256 * 2; int waves = blocks * 4; unsigned long long* cyc; double* sink;
    (void)hipMalloc(&cyc, waves * 8); (void)hipMalloc(&sink, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, 100, cyc, sink); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, 40000, cyc, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(waves); (void)hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost); std::sort(hc.begin(), hc.end());
    printf("4x4x4_4b: %.3f ms, %.1f cyc/MFMA/wave (2 waves/SIMD), %.1f TFLOP/s\n", ms, hc[waves / 2] / (40000.0 * 8), waves * 40000.0 * 8 * 512 / ms / 1e9);
    return 0;
}
