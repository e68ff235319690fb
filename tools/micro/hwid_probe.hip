// Where do the waves of co-resident 4-wave workgroups land?  512 workgroups of 256 threads with 80 KB
// of LDS each (two per CU, as the four-slot solve kernel at N = 40) record HW_ID / XCC_ID of each wave.
// Question: do the "wave 0"s of the two workgroups of a CU share a SIMD?
//   hipcc --offload-arch=gfx950 -O2 -o hwid_probe hwid_probe.hip && ./hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // HW_REG_XCC_ID[3:0]
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    // stay resident for a while so that all 512 workgroups are placed together
    double a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
    lds[threadIdx.x] = a;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 255] == 12345.0) out[0] = 0;
}
int main()
{
    const int G = 512;
    unsigned *d;
    hipMalloc(&d, G * 4 * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 80 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // key: physical CU = (xcc, se, sh, cu)
    std::map<unsigned, std::vector<int>> cu_wgs;
    int same_simd = 0, pairs = 0, distinct4 = 0;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 8], xcc = h[b * 8 + 1];
        const unsigned key = (xcc << 16) | (hw & 0xff00);
        cu_wgs[key].push_back(b);
        unsigned m = 0;
        for (int w = 0; w < 4; ++w) m |= 1u << ((h[(b * 4 + w) * 2] >> 4) & 3);
        distinct4 += m == 0xf;
    }
    for (auto &kv : cu_wgs) {
        if (kv.second.size() == 2) {
            ++pairs;
            const int a = kv.second[0], b = kv.second[1];
            same_simd += ((h[a * 8] >> 4) & 3) == ((h[b * 8] >> 4) & 3);
        }
    }
    printf("CUs seen %zu, with two workgroups %d, wave 0 of both on the same SIMD %d, workgroups with 4 distinct SIMDs %d of %d\n",
           cu_wgs.size(), pairs, same_simd, distinct4, G);
    for (int b = 0; b < 6; ++b) {
        printf("wg %d: xcc %u cu %u se %u simd of waves", b, h[b * 8 + 1], (h[b * 8] >> 8) & 0xf, (h[b * 8] >> 13) & 7);
        for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
        printf("\n");
    }
    int shown = 0;
    for (auto &kv : cu_wgs) {
        if (shown++ >= 6) break;
        printf("cu key %06x:", kv.first);
        for (int b : kv.second) printf(" wg %d (wave0 simd %u)", b, (h[b * 8] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
