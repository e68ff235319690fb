// Does an LDS read cost less when part of the wave is masked off (EXEC)?  Eight waves per CU issue
// ds_read_b128 / ds_read_b64 back to back from conflict-free addresses; the active-lane count K is 64,
// 48, 32 or 16 (lanes 0..K-1).  Reported: LDS instructions per CU per microsecond.
//   hipcc --offload-arch=gfx950 -O2 -o lds_exec_rate lds_exec_rate.hip && ./lds_exec_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WIDE>
__global__ __launch_bounds__(64) void k(double *out, int iters, int active)
{
    __shared__ double lds[64 * 2 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = i;
    __syncthreads();
    double a0 = 0, a1 = 0, b0 = 0, b1 = 0;   // (a0,a1) and (b0,b1) are the 16-byte targets
    if (lane < active) {
        const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) double *)(lds + lane * (WIDE ? 2 : 1));
        for (int i = 0; i < iters; ++i) {
            // eight reads in flight per wait: the LDS pipe, not the round trip, sets the pace
            if (WIDE)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048\n\t"
                             "ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048\n\t"
                             "ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048\n\t"
                             "ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(*(double2 *)&a0), "=&v"(*(double2 *)&b0) : "v"(addr) : "memory");
            else
                asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:2048\n\t"
                             "ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:2048\n\t"
                             "ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:2048\n\t"
                             "ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(a0), "=&v"(b0) : "v"(addr) : "memory");
        }
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + b0 + b1;
}
int main()
{
    double *d;
    hipMalloc(&d, 2048 * 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * 8;
    for (int wide = 0; wide < 2; ++wide)
        for (int act : {64, 48, 32, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (wide) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, d, iters, act);
                else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, iters, act);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s active %2d: %.3f ms, %.1f LDS instr per CU per us, %.2f cycles per instr per CU at 2.4 GHz\n",
                   wide ? "b128" : "b64 ", act, ms, 8.0 * 8 * iters / (ms * 1e3), ms * 1e3 * 2400 / (8.0 * 8 * iters));
        }
    return 0;
}
