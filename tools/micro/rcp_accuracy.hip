// Micro-test: relative error of v_rcp_f64 with 0, 1, 2 Newton steps on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    double y = __builtin_amdgcn_rcp(a);
    r0[i] = y;
    y = fma(fma(-a, y, 1.0), y, y);
    r1[i] = y;
    y = fma(fma(-a, y, 1.0), y, y);
    r2[i] = y;
}

int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = ldexp(1.0 + u, (int)(s % 80) - 40) * ((s & 1) ? 1.0 : -1.0);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)h[i];
        e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("max relative error: rcp %.3e, +1 Newton %.3e, +2 Newton %.3e (eps = %.3e)\n", e0, e1, e2, 2.2e-16);
    return 0;
}
