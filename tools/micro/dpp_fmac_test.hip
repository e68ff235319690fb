// Micro-test: does v_fmac_f64 with DPP row_newbcast do what the ISA text says on gfx950, and what
// does a dependent chain of them cost against the v_readlane formulation?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

template <int C>
__device__ inline double fmac_bcast(double acc, double src, double coef)
{
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(coef), "n"(C));
    return acc;
}

__device__ inline double readlane_d(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__global__ void check(double *out, const double *in)
{
    const double x = in[threadIdx.x], c = in[64 + threadIdx.x], a = in[128 + threadIdx.x];
    double r = fmac_bcast<3>(a, x, c);
    r = fmac_bcast<11>(r, x, c);
    out[threadIdx.x] = r;
}

// dependent 6x6 mat-vec recursion, 8 fmacs per step
__global__ void chain_dpp(double *out, const double *in, int steps)
{
    double x = in[threadIdx.x];
    double c[8];
    for (int i = 0; i < 8; ++i) c[i] = in[64 + ((threadIdx.x + i) & 63)] * 0.1;
    for (int s = 0; s < steps; ++s) {
        double a0 = c[7], a1 = 0.0;
        a0 = fmac_bcast<0>(a0, x, c[0]); a1 = fmac_bcast<1>(a1, x, c[1]);
        a0 = fmac_bcast<2>(a0, x, c[2]); a1 = fmac_bcast<3>(a1, x, c[3]);
        a0 = fmac_bcast<4>(a0, x, c[4]); a1 = fmac_bcast<5>(a1, x, c[5]);
        double y = a0 + a1;
        y = fmac_bcast<6>(y, y, c[6]); 
        x = y;
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

__global__ void chain_readlane(double *out, const double *in, int steps)
{
    double x = in[threadIdx.x];
    double c[8];
    for (int i = 0; i < 8; ++i) c[i] = in[64 + ((threadIdx.x + i) & 63)] * 0.1;
    double x0 = readlane_d(x, 0), x1 = readlane_d(x, 1), x2 = readlane_d(x, 2), x3 = readlane_d(x, 3),
           x4 = readlane_d(x, 4), x5 = readlane_d(x, 5);
    for (int s = 0; s < steps; ++s) {
        double a0 = fma(x0, c[0], c[7]), a1 = x1 * c[1];
        a0 = fma(x2, c[2], a0); a1 = fma(x3, c[3], a1);
        a0 = fma(x4, c[4], a0); a1 = fma(x5, c[5], a1);
        double y = a0 + a1;
        const double y6 = readlane_d(y, 6);
        y = fma(y6, c[6], y);
        x0 = readlane_d(y, 0); x1 = readlane_d(y, 1); x2 = readlane_d(y, 2);
        x3 = readlane_d(y, 3); x4 = readlane_d(y, 4); x5 = readlane_d(y, 5);
        x = y;
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

int main()
{
    double h[192], o[64];
    for (int i = 0; i < 192; ++i) h[i] = 0.01 * (i + 1) + (i % 7) * 0.3;
    double *din, *dout;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, 2048 * 64 * sizeof(double));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dout, din);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15;
        const double want = fma(h[row + 11], h[64 + l], fma(h[row + 3], h[64 + l], h[128 + l]));
        if (o[l] != want) { ++bad; if (bad < 5) printf("lane %d got %.17g want %.17g\n", l, o[l], want); }
    }
    printf("check: %d bad lanes\n", bad);
    const int steps = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {1, 2}) {
        for (int v = 0; v < 2; ++v) {
            const int blocks = 1024 * waves;   // waves per SIMD resident: 256 CUs x 4 SIMDs = 1024
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(chain_dpp, dim3(blocks), dim3(64), 0, 0, dout, din, steps);
                else hipLaunchKernelGGL(chain_readlane, dim3(blocks), dim3(64), 0, 0, dout, din, steps);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s waves/SIMD=%d: %.3f ms, %.1f ns/step\n", v == 0 ? "dpp     " : "readlane", waves, ms,
                   ms * 1e6 / steps);
        }
    }
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("sink %g\n", o[3]);
    return bad != 0;
}
