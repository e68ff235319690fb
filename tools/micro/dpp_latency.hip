// Micro-benchmark: dependent-issue latency of v_fma_f64, v_fmac_f64_dpp (with/without s_nop), 1 wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int C, int NOP>
__device__ inline double fmac_bcast(double acc, double src, double coef)
{
    if (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(coef), "n"(C));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(coef), "n"(C));
    return acc;
}

// mode 0: plain fma chain (8 dependent); 1: dpp chain acc-dependent (src fixed) with nop; 2: same without nop;
// 3: dpp chain where src is the previous result (src-dependent) with nop; 4: same without nop
template <int MODE>
__global__ void chain(double *out, const double *in, int steps)
{
    double x = in[threadIdx.x], c = in[64 + threadIdx.x] * 1e-3, a = in[128 + threadIdx.x];
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(c)); }
        } else if (MODE == 1) {
            a = fmac_bcast<0, 1>(a, x, c); a = fmac_bcast<1, 1>(a, x, c); a = fmac_bcast<2, 1>(a, x, c); a = fmac_bcast<3, 1>(a, x, c);
            a = fmac_bcast<4, 1>(a, x, c); a = fmac_bcast<5, 1>(a, x, c); a = fmac_bcast<6, 1>(a, x, c); a = fmac_bcast<7, 1>(a, x, c);
        } else if (MODE == 2) {
            a = fmac_bcast<0, 0>(a, x, c); a = fmac_bcast<1, 0>(a, x, c); a = fmac_bcast<2, 0>(a, x, c); a = fmac_bcast<3, 0>(a, x, c);
            a = fmac_bcast<4, 0>(a, x, c); a = fmac_bcast<5, 0>(a, x, c); a = fmac_bcast<6, 0>(a, x, c); a = fmac_bcast<7, 0>(a, x, c);
        } else if (MODE == 3) {
            a = fmac_bcast<0, 1>(a, a, c); a = fmac_bcast<1, 1>(a, a, c); a = fmac_bcast<2, 1>(a, a, c); a = fmac_bcast<3, 1>(a, a, c);
            a = fmac_bcast<4, 1>(a, a, c); a = fmac_bcast<5, 1>(a, a, c); a = fmac_bcast<6, 1>(a, a, c); a = fmac_bcast<7, 1>(a, a, c);
        } else {
            a = fmac_bcast<0, 0>(a, a, c); a = fmac_bcast<1, 0>(a, a, c); a = fmac_bcast<2, 0>(a, a, c); a = fmac_bcast<3, 0>(a, a, c);
            a = fmac_bcast<4, 0>(a, a, c); a = fmac_bcast<5, 0>(a, a, c); a = fmac_bcast<6, 0>(a, a, c); a = fmac_bcast<7, 0>(a, a, c);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}

template <int MODE>
void run(double *dout, const double *din, int waves)
{
    const int steps = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain<MODE>, dim3(1024 * waves), dim3(64), 0, 0, dout, din, steps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("mode %d waves/SIMD %d: %.2f ns per op (%.1f cycles at 2.4 GHz)\n", MODE, waves, ms * 1e6 / steps / 8,
           ms * 1e6 / steps / 8 * 2.4);
}

int main()
{
    double h[192];
    for (int i = 0; i < 192; ++i) h[i] = 0.01 * (i + 1);
    double *din, *dout;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, 4096 * 64 * sizeof(double));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<0>(dout, din, w); run<1>(dout, din, w); run<2>(dout, din, w); run<3>(dout, din, w); run<4>(dout, din, w);
    }
    // does the src-dependent chain without s_nop give the same numbers as with it?
    double a[64], b[64];
    hipLaunchKernelGGL(chain<3>, dim3(1), dim3(64), 0, 0, dout, din, 3);
    hipMemcpy(a, dout, sizeof(a), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(chain<4>, dim3(1), dim3(64), 0, 0, dout, din, 3);
    hipMemcpy(b, dout, sizeof(b), hipMemcpyDeviceToHost);
    int diff = 0;
    for (int i = 0; i < 64; ++i) diff += a[i] != b[i];
    printf("src-dependent chain: %d lanes differ without the s_nop\n", diff);
    return 0;
}
