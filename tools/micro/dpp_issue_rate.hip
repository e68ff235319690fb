// Issue cost of v_fmac_f64 with and without a DPP row broadcast, independent and dependent chains,
// one wave alone on a SIMD (cycles per instruction from s_memtime).  Tools only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(double *out, long long *cyc, const double *in)
{
    double a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5;
    double s = in[64 + threadIdx.x] * 1e-3, c = in[128 + threadIdx.x] * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {        // six independent chains, DPP
            asm volatile(REP16(
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %2, %6, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %3, %6, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %4, %6, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %5, %6, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(s), "v"(c));
        } else if (MODE == 1) { // six independent chains, plain
            asm volatile(REP16(
                "v_fmac_f64 %0, %6, %7\n\tv_fmac_f64 %1, %6, %7\n\tv_fmac_f64 %2, %6, %7\n\t"
                "v_fmac_f64 %3, %6, %7\n\tv_fmac_f64 %4, %6, %7\n\tv_fmac_f64 %5, %6, %7\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(s), "v"(c));
        } else if (MODE == 2) { // one dependent chain, DPP (accumulator dependency only)
            asm volatile(REP16(
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(s), "v"(c));
        } else if (MODE == 3) { // one dependent chain, plain
            asm volatile(REP16(
                "v_fmac_f64 %0, %6, %7\n\tv_fmac_f64 %0, %6, %7\n\tv_fmac_f64 %0, %6, %7\n\t"
                "v_fmac_f64 %0, %6, %7\n\tv_fmac_f64 %0, %6, %7\n\tv_fmac_f64 %0, %6, %7\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(s), "v"(c));
        } else if (MODE == 4) { // two alternating chains, DPP
            asm volatile(REP16(
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, %6, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %6, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, %6, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(s), "v"(c));
        } else if (MODE == 5) { // fp32 plain for scale
            float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, fs = (float)s, fc = (float)c;
            asm volatile(REP16(
                "v_fmac_f32 %0, %6, %7\n\tv_fmac_f32 %1, %6, %7\n\tv_fmac_f32 %2, %6, %7\n\t"
                "v_fmac_f32 %3, %6, %7\n\tv_fmac_f32 %4, %6, %7\n\tv_fmac_f32 %5, %6, %7\n\t")
                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(fs), "v"(fc));
            a0 += f0 + f1 + f2 + f3 + f4 + f5;
        } else if (MODE == 6) { // ds_read_b64 issue, independent, waited every 12
            unsigned addr = threadIdx.x * 8;
            asm volatile(REP16(
                "ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:512\n\tds_read_b64 %2, %6 offset:1024\n\t"
                "ds_read_b64 %3, %6 offset:1536\n\tds_read_b64 %4, %6 offset:2048\n\tds_read_b64 %5, %6 offset:2560\n\t")
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : "v"(addr), "v"(c));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, int blocks, double *dout, long long *dcyc, double *din)
{
    long long h[4096];
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 4096, 0, dout, dcyc, din);
    hipMemcpy(h, dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += h[i];
    printf("%-34s blocks=%5d  %.2f cycles/instr (s_memtime ticks, 100 MHz x ratio -> see note)\n", name, blocks, s / blocks / (64.0 * 96));
}
int main()
{
    double h[192];
    for (int i = 0; i < 192; ++i) h[i] = 0.01 * (i + 1);
    double *din, *dout; long long *dcyc;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 4096 * 64 * sizeof(double)); hipMalloc(&dcyc, 4096 * sizeof(long long));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    for (int blocks : {1, 1024, 2048}) {
        run<0>("dpp f64 fmac, 6 independent", blocks, dout, dcyc, din);
        run<1>("plain f64 fmac, 6 independent", blocks, dout, dcyc, din);
        run<2>("dpp f64 fmac, dependent chain", blocks, dout, dcyc, din);
        run<3>("plain f64 fmac, dependent chain", blocks, dout, dcyc, din);
        run<4>("dpp f64 fmac, 2 chains", blocks, dout, dcyc, din);
        run<5>("plain f32 fmac, 6 independent", blocks, dout, dcyc, din);
        run<6>("ds_read_b64 x6 + wait", blocks, dout, dcyc, din);
    }
    return 0;
}
