// FP64 VALU FMA rate, alone and next to FP64 MFMA waves (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));

// mode 0: all waves VALU; mode 1: all waves MFMA; mode 2: waves with (wave&1)==0 MFMA, others VALU
__global__ __launch_bounds__(512) void probe(int mode, int iters, const double* coef, unsigned long long* cyc, int* kind, double* sink) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (mode == 1) || (mode == 2 && (wave & 1) == 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    if (do_mfma) {
        v4d acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
        double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        // 32 independent accumulators, multiplier a per lane, wave-uniform coefficients (SGPR operands)
        double acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = 0.0;
        double a = 1.0 + 1e-9 * threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            const double* c = coef + (it & 7) * 32;
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = fma(a, c[i], acc[i]);
            a += 1e-12;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc[i];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x / 64) + wave;
        cyc[w] = t1 - t0;
        kind[w] = do_mfma;
    }
}

void run(int mode, int blocks, int threads, int iters) {
    int waves = blocks * threads / 64;
    unsigned long long* cyc; int* kind; double *sink, *coef;
    (void)hipMalloc(&cyc, waves * 8); (void)hipMalloc(&kind, waves * 4); (void)hipMalloc(&sink, 8); (void)hipMalloc(&coef, 256 * 8);
    std::vector<double> hcoef(256); for (int i = 0; i < 256; ++i) hcoef[i] = 1.0 + i * 1e-6;
    (void)hipMemcpy(coef, hcoef.data(), 256 * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, mode, 100, coef, cyc, kind, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, mode, iters, coef, cyc, kind, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> hc(waves); std::vector<int> hk(waves);
    (void)hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hk.data(), kind, waves * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> cm, cv;
    for (int i = 0; i < waves; ++i) (hk[i] ? cm : cv).push_back(hc[i]);
    double fl = 0;
    printf("mode=%d blocks=%d thr=%d: %.3f ms", mode, blocks, threads, ms);
    if (!cm.empty()) { std::sort(cm.begin(), cm.end()); double c = cm[cm.size() / 2]; fl += cm.size() * (double)iters * 4 * 2048;
        printf(" | MFMA waves %zu: %.1f cyc/MFMA", cm.size(), c / (iters * 4.0)); }
    if (!cv.empty()) { std::sort(cv.begin(), cv.end()); double c = cv[cv.size() / 2]; fl += cv.size() * (double)iters * 32 * 128;
        printf(" | VALU waves %zu: %.2f cyc/DFMA", cv.size(), c / (iters * 32.0)); }
    printf(" | total %.1f TFLOP/s\n", fl / ms / 1e9);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    const int it = 20000;
    run(0, cus, 256, it);   // VALU only, 1 wave/SIMD
    run(0, cus, 512, it);   // VALU only, 2 waves/SIMD
    run(1, cus, 256, it);   // MFMA only, 1 wave/SIMD
    run(1, cus, 512, it);   // MFMA only, 2 waves/SIMD
    run(2, cus, 512, it);   // mixed: 1 MFMA + 1 VALU wave per SIMD
    run(2, cus * 2, 512, it);  // mixed: 2 + 2 per SIMD
    return 0;
}
