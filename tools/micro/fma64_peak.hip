// Micro-benchmark: fp64 vector FMA throughput of the whole chip (the denominator of bench.py's
// compute roofline: 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz).  Each wave
// runs 8 independent v_fma_f64 chains; 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) fma_chain(double *out, int iters)
{
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double x = 0.999999, y = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
            a4 = fma(a4, x, y); a5 = fma(a5, x, y); a6 = fma(a6, x, y); a7 = fma(a7, x, y);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    const int blocks = 256 * 8, threads = 256, iters = 4000;
    double *d;
    if (hipMalloc(&d, sizeof(double) * blocks * threads) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma_chain, dim3(blocks), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2.0 * 64.0 * iters * (double)blocks * threads;
    printf("fp64 v_fma_f64: %.1f TFLOP/s (%.3f ms for %.3g flop)\n", flop / (best * 1e-3) / 1e12, best, flop);
    hipFree(d);
    return 0;
}
