// Micro-benchmark: LDS pipe cycles of ds_read_b64 / ds_read_b128 as a function of the number of
// active lanes (does an EXEC-masked wave pay for 64 lanes?).  8 waves per CU, LDS-saturating.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WIDTH>
__global__ void __launch_bounds__(64) rd(double *out, int active, int iters)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = i;
    __syncthreads();
    double acc = 0.0;
    if ((int)threadIdx.x < active) {
        typedef __attribute__((address_space(3))) double lds_d;
        unsigned a = (unsigned)(unsigned long long)(lds_d *)(lds + 2 * (threadIdx.x & 7));
        for (int it = 0; it < iters; ++it) {
            if (WIDTH == 8) {
                double v0, v1, v2, v3, v4, v5, v6, v7;
                asm volatile(
                    "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:128\n\tds_read_b64 %2, %8 offset:256\n\t"
                    "ds_read_b64 %3, %8 offset:384\n\tds_read_b64 %4, %8 offset:512\n\tds_read_b64 %5, %8 offset:640\n\t"
                    "ds_read_b64 %6, %8 offset:768\n\tds_read_b64 %7, %8 offset:896\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                    : "v"(a) : "memory");
                acc += v0 + v7;
            } else {
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 v0, v1, v2, v3, v4, v5, v6, v7;
                asm volatile(
                    "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:128\n\tds_read_b128 %2, %8 offset:256\n\t"
                    "ds_read_b128 %3, %8 offset:384\n\tds_read_b128 %4, %8 offset:512\n\tds_read_b128 %5, %8 offset:640\n\t"
                    "ds_read_b128 %6, %8 offset:768\n\tds_read_b128 %7, %8 offset:896\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                    : "v"(a) : "memory");
                acc += v0.x + v7.y;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
    double *dout;
    hipMalloc(&dout, 4096 * 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int width : {8, 16})
        for (int active : {64, 48, 32, 16, 8}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (width == 8) hipLaunchKernelGGL(rd<8>, dim3(2048), dim3(64), 16384, 0, dout, active, iters);
                else hipLaunchKernelGGL(rd<16>, dim3(2048), dim3(64), 16384, 0, dout, active, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            // 8 waves per CU, each iters*8 instructions
            const double ns_per_instr_cu = ms * 1e6 / (8.0 * iters * 8.0);
            printf("ds_read_b%d active %2d lanes: %.2f ns per instruction per CU (%.1f clk at 2.4 GHz)\n", width * 8,
                   active, ns_per_instr_cu, ns_per_instr_cu * 2.4);
        }
    return 0;
}
