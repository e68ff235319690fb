#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, const double *in)
{
    double x = in[threadIdx.x];
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(x));
    out[threadIdx.x] = r;
    double acc = 0.5, one = 1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(one));
    out[64 + threadIdx.x] = acc;
}
int main()
{
    double h[64], o[128];
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 0.25;
    double *din, *dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, din);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) {
        if (o[l] != h[(l & ~15) + 9]) { if (bad < 4) printf("mov lane %d got %g want %g\n", l, o[l], h[(l & ~15) + 9]); ++bad; }
        if (o[64 + l] != 0.5 - h[(l & ~15) + 10]) { if (bad2 < 4) printf("fmacneg lane %d got %g want %g\n", l, o[64 + l], 0.5 - h[(l & ~15) + 10]); ++bad2; }
    }
    printf("mov_b64_dpp bad %d, fmac neg bad %d\n", bad, bad2);
    return 0;
}
