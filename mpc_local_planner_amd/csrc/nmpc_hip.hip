// nmpc_hip.hip -- gfx950 kernels and the C ABI (include/nmpc.h) of the batched
// path-tracking NMPC solver.  One 64-lane wavefront (= one 64-thread workgroup)
// per OCP instance; see nmpc_wave_solver.h for the in-wave work mapping.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#define NMPC_DEV __device__ __forceinline__
#include "nmpc_wave_solver.h"
#include "nmpc_quad_solver.h"
#include "nmpc_params_file.h"
#include "nmpc_frontend.h"

#include "../../include/nmpc.h"

namespace {

// ------------------------------------------------------------------ device side

// NW = waves per OCP instance (= per workgroup): 1 for N <= 63, 2 for N <= 127.  The sweeps always
// run on the first wave; the other wave only takes part in the stage-layout work.
// SQP = full SQP mode (the RTI instantiation carries none of the KKT-residual / costate code).
#ifdef NMPC_WAVE_TIMELINE
__device__ unsigned long long g_wave_tl[24];
#endif
template <int NW, bool SQP, bool COMPACT = false>
struct GpuWave {
    using Layout = typename std::conditional<COMPACT, nmpcw::Layout48, nmpcw::Layout54>::type;
    static constexpr bool kSqp = SQP;
    static constexpr bool kNeighbourLds = NW > 1;   // x_{k+1} of the next node: LDS exchange instead of a lane shuffle
    double *lds_;
    double *pol_;      // NW > 1: [2][4] reduction slots
    int flip_ = 0;
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int lane() const { return (int)threadIdx.x & 63; }
    __device__ __forceinline__ bool sweep_wave() const { return NW == 1 || threadIdx.x < 64; }
    __device__ __forceinline__ bool uniform_flag(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }   // b is the same on every lane
#ifdef NMPC_WAVE_TIMELINE
    // Debug build (tools/wave_timeline.py): shader-clock ticks of the sweeping wave per phase, accumulated in
    // the last 24 slots of the instance's exchange area by thread 0 and added to g_wave_tl when the instance ends.
    // State in a scalar register and LDS only (the kernel has no vector register to spare).
    int tl_off = 0;
    __device__ __forceinline__ unsigned long long *tl_ptr() { return (unsigned long long *)lds_ + __builtin_amdgcn_readfirstlane(tl_off); }
    __device__ __forceinline__ void stamp_init(double *area)
    {
        tl_off = __builtin_amdgcn_readfirstlane((int)(area - lds_));
        if (threadIdx.x < 23) tl_ptr()[threadIdx.x] = 0ull;
        if (threadIdx.x == 23) tl_ptr()[23] = (unsigned long long)__builtin_readcyclecounter();    // previous stamp
    }
    __device__ __forceinline__ void stamp(int id)
    {
        if (threadIdx.x == 0) {
            unsigned long long *tl = tl_ptr();
            const unsigned long long t = (unsigned long long)__builtin_readcyclecounter();
            tl[id] += t - tl[23];
            tl[23] = (unsigned long long)__builtin_readcyclecounter();
        }
    }
    __device__ __forceinline__ void tick(int id) { if (threadIdx.x == 0) tl_ptr()[id] += 1ull; }
    __device__ __forceinline__ void stamp_flush();
#else
    __device__ __forceinline__ void stamp(int) {}
    __device__ __forceinline__ void tick(int) {}
    __device__ __forceinline__ void stamp_init(double *) {}
    __device__ __forceinline__ void stamp_flush() {}
#endif
    __device__ __forceinline__ double *lds() { return lds_; }
    // value the optimiser must treat as unknown: keeps per-sweep address set-up inside the
    // interior-point loop instead of hoisted into ~45 live registers
    __device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
    // a rounded fp64 value the compiler may not fold into the arithmetic around it (no contraction
    // across this point: the result is the same in every instantiation of the kernel)
    __device__ __forceinline__ double rounded(double v) { asm volatile("" : "+v"(v)); return v; }
    // Exchange through LDS between lanes of ONE wave: the LDS executes a wave's instructions in
    // issue order, so a write followed by a read needs no s_waitcnt and no s_barrier, only a
    // fence that keeps the compiler from moving LDS accesses across the exchange point.
    __device__ __forceinline__ void wsync()
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // exchange between all threads of the instance
    __device__ __forceinline__ void sync()
    {
        if (NW == 1) wsync(); else __syncthreads();
    }
    __device__ __forceinline__ void sync_lockstep() { wsync(); }
    // ---- vector sweeps.  A stage is ONE asm statement: the loads of the NEXT stage's data are
    // issued first, then the multiply-add chain of the current stage runs on registers, then the
    // result is stored and the loads are waited for.  v_fmac_f64 with a DPP row broadcast
    // (row_newbcast:C, the only DPP control fp64 has) multiplies by lane C of the caller's 16-lane
    // row, so the recursion vector never leaves the VALU.  Hand-written because (a) the compiler
    // merges the strided 8-byte loads into ds_read2_b64 (8 LDS cycles per pair instead of 2+2),
    // (b) it does not fold a 64-bit DPP move into the multiply-add, (c) a load that is waited for
    // in a later statement is not something C++ can express.  The closing wait leaves the stage's own
    // store in flight (LDS operations of a wave complete in order, so lgkmcnt(1) covers the loads); the
    // compiler's later waits are only more conservative for it, and sweep_drain() ends each sweep.
    // A register read through DPP needs two wait states after the VALU write of it (the hardware
    // does not interlock: a build without them failed tests/test_gpu_cpp_mirror.py); the compiler
    // cannot see into the asm, so each chain provides them itself (an independent instruction + s_nop 0).
    static __device__ __forceinline__ unsigned lds_addr(const double *p)
    {
        typedef __attribute__((address_space(3))) const double lds_cd;
        return (unsigned)(unsigned long long)(lds_cd *)p;
    }
    // d[0..5] = p1[0,6,..,30], d[6,7] = p3[0,2], d[8] = p2[0]
    __device__ __forceinline__ void fwd_fetch(const double *p1, const double *p3, const double *p2, double (&d)[9])
    {
        asm volatile(
            "ds_read_b64 %8, %11\n\t"
            "ds_read_b64 %6, %10\n\t"
            "ds_read_b64 %7, %10 offset:16\n\t"
            "ds_read_b64 %2, %9 offset:96\n\t"
            "ds_read_b64 %3, %9 offset:144\n\t"
            "ds_read_b64 %4, %9 offset:192\n\t"
            "ds_read_b64 %5, %9 offset:240\n\t"
            "ds_read_b64 %0, %9\n\t"
            "ds_read_b64 %1, %9 offset:48\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]),
              "=&v"(d[7]), "=&v"(d[8])
            : "v"(lds_addr(p1)), "v"(lds_addr(p3)), "v"(lds_addr(p2))
            : "memory");
    }
    // EXEC-masking the lanes with constant operands out of the per-stage loads was tried (it removes
    // their bank conflicts with the stage data: SQ_LDS_BANK_CONFLICT 15.1k -> 8.2k per solve) and lost
    // 3 % overall: each s_mov exec ahead of a vector instruction costs more than the conflict cycles.
    __device__ __forceinline__ void fwd_stage(double (&cur)[9], double (&nxt)[9], const double *p1,
                                              const double *p3, const double *p2, double *sw, double &val)
    {
        double t;
        asm volatile(
            "ds_read_b64 %[n8], %[a2]\n\t"
            "ds_read_b64 %[n6], %[a3]\n\t"
            "ds_read_b64 %[n7], %[a3] offset:16\n\t"
            "ds_read_b64 %[n2], %[a1] offset:96\n\t"
            "ds_read_b64 %[n3], %[a1] offset:144\n\t"
            "ds_read_b64 %[n4], %[a1] offset:192\n\t"
            "ds_read_b64 %[n5], %[a1] offset:240\n\t"
            "ds_read_b64 %[n0], %[a1]\n\t"
            "ds_read_b64 %[n1], %[a1] offset:48\n\t"
            "v_fmac_f64_dpp %[iv], %[val], %[e0] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b64 %[t], 0\n\t"
            "v_fmac_f64_dpp %[t], %[val], %[e1] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[iv], %[val], %[c2] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[t], %[val], %[c3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[iv], %[val], %[c4] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[t], %[val], %[c5] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f64 %[val], %[iv], %[t]\n\t"     // level-1 result: what the two DPP reads below take
            "v_add_f64 %[t], %[iv], %[t]\n\t"       // the same sum again as the accumulator: with the s_nop
            "s_nop 0\n\t"                           // that is the two wait states between write and DPP read
            "v_fmac_f64_dpp %[t], %[val], %[c0] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[t], %[val], %[c1] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "ds_write_b64 %[sw], %[t]\n\t"
            "s_waitcnt lgkmcnt(1)"
            : [n0] "=&v"(nxt[0]), [n1] "=&v"(nxt[1]), [n2] "=&v"(nxt[2]), [n3] "=&v"(nxt[3]), [n4] "=&v"(nxt[4]),
              [n5] "=&v"(nxt[5]), [n6] "=&v"(nxt[6]), [n7] "=&v"(nxt[7]), [n8] "=&v"(nxt[8]), [t] "=&v"(t),
              [val] "+v"(val), [iv] "+v"(cur[8])
            : [c0] "v"(cur[0]), [c1] "v"(cur[1]), [c2] "v"(cur[2]), [c3] "v"(cur[3]), [c4] "v"(cur[4]),
              [c5] "v"(cur[5]), [e0] "v"(cur[6]), [e1] "v"(cur[7]), [a1] "v"(lds_addr(p1)), [a3] "v"(lds_addr(p3)),
              [a2] "v"(lds_addr(p2)), [sw] "v"(lds_addr(sw))
            : "memory");
        val = t;
    }
    // d[0..2] = tv[0..2], d[3] = dg[0], d[4,5] = ab[0,1]
    __device__ __forceinline__ void bwd_fetch(const double *tv, const double *dg, const double *ab, double (&d)[6])
    {
        asm volatile(
            "ds_read_b64 %0, %6\n\t"
            "ds_read_b64 %1, %6 offset:8\n\t"
            "ds_read_b64 %2, %6 offset:16\n\t"
            "ds_read_b64 %3, %7\n\t"
            "ds_read_b64 %4, %8\n\t"
            "ds_read_b64 %5, %8 offset:8\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
            : "v"(lds_addr(tv)), "v"(lds_addr(dg)), "v"(lds_addr(ab))
            : "memory");
    }
    __device__ __forceinline__ void bwd_stage(double (&cur)[6], double (&nxt)[6], const double *tv, const double *dg,
                                              const double *ab, double *kfw, double cb0, double cb1, double cb2,
                                              double keep, double &res)
    {
        double t;
        asm volatile(
            "ds_read_b64 %[n0], %[atv]\n\t"
            "ds_read_b64 %[n1], %[atv] offset:8\n\t"
            "ds_read_b64 %[n2], %[atv] offset:16\n\t"
            "ds_read_b64 %[n3], %[adg]\n\t"
            "ds_read_b64 %[n4], %[aab]\n\t"
            "ds_read_b64 %[n5], %[aab] offset:8\n\t"
            "v_fmac_f64_dpp %[g], %[res], %[t0] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b64 %[t], 0\n\t"
            "v_fmac_f64_dpp %[t], %[res], %[t1] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[g], %[res], %[t2] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[t], %[res], %[cb0] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[g], %[res], %[cb1] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[t], %[res], %[cb2] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f64 %[g], %[g], %[t]\n\t"
            "v_mul_f64 %[res], %[keep], %[g]\n\t"
            "s_nop 0\n\t"
            "v_fmac_f64_dpp %[res], %[g], %[ka] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[res], %[g], %[kb] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "ds_write_b64 %[kfw], %[res]\n\t"
            "s_waitcnt lgkmcnt(1)"
            : [n0] "=&v"(nxt[0]), [n1] "=&v"(nxt[1]), [n2] "=&v"(nxt[2]), [n3] "=&v"(nxt[3]), [n4] "=&v"(nxt[4]),
              [n5] "=&v"(nxt[5]), [t] "=&v"(t), [res] "+v"(res), [g] "+v"(cur[3])
            : [t0] "v"(cur[0]), [t1] "v"(cur[1]), [t2] "v"(cur[2]), [ka] "v"(cur[4]), [kb] "v"(cur[5]),
              [cb0] "v"(cb0), [cb1] "v"(cb1), [cb2] "v"(cb2), [keep] "v"(keep), [atv] "v"(lds_addr(tv)),
              [adg] "v"(lds_addr(dg)), [aab] "v"(lds_addr(ab)), [kfw] "v"(lds_addr(kfw))
            : "memory");
    }
    // the same two with the pair (ab[0], ab2[0]) not necessarily adjacent (Layout48: rows of -inv(Huu) in spare slots)
    __device__ __forceinline__ void bwd_fetch2(const double *tv, const double *dg, const double *ab, const double *ab2,
                                               double (&d)[6])
    {
        asm volatile(
            "ds_read_b64 %0, %6\n\t"
            "ds_read_b64 %1, %6 offset:8\n\t"
            "ds_read_b64 %2, %6 offset:16\n\t"
            "ds_read_b64 %3, %7\n\t"
            "ds_read_b64 %4, %8\n\t"
            "ds_read_b64 %5, %9\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
            : "v"(lds_addr(tv)), "v"(lds_addr(dg)), "v"(lds_addr(ab)), "v"(lds_addr(ab2))
            : "memory");
    }
    __device__ __forceinline__ void bwd_stage2(double (&cur)[6], double (&nxt)[6], const double *tv, const double *dg,
                                               const double *ab, const double *ab2, double *kfw, double cb0, double cb1,
                                               double cb2, double keep, double &res)
    {
        double t;
        asm volatile(
            "ds_read_b64 %[n0], %[atv]\n\t"
            "ds_read_b64 %[n1], %[atv] offset:8\n\t"
            "ds_read_b64 %[n2], %[atv] offset:16\n\t"
            "ds_read_b64 %[n3], %[adg]\n\t"
            "ds_read_b64 %[n4], %[aab]\n\t"
            "ds_read_b64 %[n5], %[aab2]\n\t"
            "v_fmac_f64_dpp %[g], %[res], %[t0] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b64 %[t], 0\n\t"
            "v_fmac_f64_dpp %[t], %[res], %[t1] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[g], %[res], %[t2] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[t], %[res], %[cb0] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[g], %[res], %[cb1] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[t], %[res], %[cb2] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f64 %[g], %[g], %[t]\n\t"
            "v_mul_f64 %[res], %[keep], %[g]\n\t"
            "s_nop 0\n\t"
            "v_fmac_f64_dpp %[res], %[g], %[ka] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[res], %[g], %[kb] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "ds_write_b64 %[kfw], %[res]\n\t"
            "s_waitcnt lgkmcnt(1)"
            : [n0] "=&v"(nxt[0]), [n1] "=&v"(nxt[1]), [n2] "=&v"(nxt[2]), [n3] "=&v"(nxt[3]), [n4] "=&v"(nxt[4]),
              [n5] "=&v"(nxt[5]), [t] "=&v"(t), [res] "+v"(res), [g] "+v"(cur[3])
            : [t0] "v"(cur[0]), [t1] "v"(cur[1]), [t2] "v"(cur[2]), [ka] "v"(cur[4]), [kb] "v"(cur[5]),
              [cb0] "v"(cb0), [cb1] "v"(cb1), [cb2] "v"(cb2), [keep] "v"(keep), [atv] "v"(lds_addr(tv)),
              [adg] "v"(lds_addr(dg)), [aab] "v"(lds_addr(ab)), [aab2] "v"(lds_addr(ab2)), [kfw] "v"(lds_addr(kfw))
            : "memory");
    }
    // issue priority of the wave while it runs the serial sweeps: the other wave of the SIMD, when it is
    // in its element-wise phase, has independent work to fill the gaps with (+0.3 % cold, +1.2 % warm)
    __device__ __forceinline__ void sweep_priority(bool on)
    {
        if (on) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    }
    // after a sweep: nothing of it is still in flight
    __device__ __forceinline__ void sweep_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    // acc + (src of lane C of the caller's 16-lane row) * coef, fused (single use outside the sweeps' asm)
    template <int C>
    __device__ __forceinline__ double fmac_row(double acc, double src, double coef)
    {
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(src), "v"(coef), "n"(C));
        return acc;
    }
    // keep the loads issued above this point together (one wait instead of one per load)
    __device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
    // a value that is the same on every lane, moved to scalar registers
    __device__ __forceinline__ double uniform(double v)
    {
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
        const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
        return __hiloint2double(hi, lo);
    }
    // value of lane `src` of the calling wave (uniform)
    __device__ __forceinline__ double readlane(double v, int src)
    {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    }
    // value of lane + 1 (the last lane gets its own); one-wave instances only -- two-wave instances
    // exchange through the LDS image (kNeighbourLds)
    __device__ __forceinline__ double shfl_down1(double v) { return __shfl_down(v, 1, 64); }
    // Reductions over all threads of the instance without LDS traffic inside a wave: four DPP
    // butterfly steps (xor 1, xor 2, mirror within 8, mirror within 16) leave each 16-lane row's
    // result in all its lanes; the four row results are read with v_readlane and combined.
    template <int CTRL>
    __device__ __forceinline__ static double dpp(double v)
    {
        // every lane has a valid source under these controls (full rows, full EXEC): no "old" value to set up
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    // partial results of the waves meet in LDS; slots alternate so one barrier per call is enough
    template <class OP>
    __device__ __forceinline__ double across_waves(double s, OP op)
    {
        if (NW == 1) return s;
        double *slot = pol_ + (flip_ & 1) * 4;
        flip_ ^= 1;
        if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = s;
        __syncthreads();
        double r = slot[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) r = op(r, slot[i]);
        return r;
    }
    __device__ __forceinline__ double sum(double v)
    {
        v += dpp<0xB1>(v);     // quad_perm [1,0,3,2]
        v += dpp<0x4E>(v);     // quad_perm [2,3,0,1]
        v += dpp<0x141>(v);    // row_half_mirror
        v += dpp<0x140>(v);    // row_mirror
        const double s = (readlane(v, 0) + readlane(v, 16)) + (readlane(v, 32) + readlane(v, 48));
        return across_waves(s, [](double a, double b) { return a + b; });
    }
    __device__ __forceinline__ double max(double v)
    {
        v = fmax(v, dpp<0xB1>(v));
        v = fmax(v, dpp<0x4E>(v));
        v = fmax(v, dpp<0x141>(v));
        v = fmax(v, dpp<0x140>(v));
        const double s = fmax(fmax(readlane(v, 0), readlane(v, 16)), fmax(readlane(v, 32), readlane(v, 48)));
        return across_waves(s, [](double a, double b) { return fmax(a, b); });
    }
    // reciprocal to ~1 ulp: hardware seed y (relative error e ~ 5e-8, tools/micro/rcp_accuracy.hip) times
    // 1 + e + e^2 (error e^3) -- three dependent multiply-adds where two Newton steps take four; no
    // div_scale/div_fixup sequence, operands here are never subnormal or huge
    __device__ __forceinline__ double rcp(double x)
    {
        const double y = __builtin_amdgcn_rcp(x);
        const double e = fma(-x, y, 1.0);
        return fma(y, fma(e, e, e), y);
    }
};

enum KernelChoice { KERNEL_AUTO = 0, KERNEL_WAVE, KERNEL_QUAD };   // see use_quad_kernel

struct KernelArgs {
    nmpcw::Problem pb;
    int B;
    const double *x0, *p, *W, *bnd;
    int w_stride, bnd_stride;   // doubles between the rows of W / bnd: 7 / 8, or 0 when all instances share one row
    double *traj, *ctrl, *cost;
    int *status, *iters, *sqp_iters;
    unsigned char *reset;
    const int *order;   // launch slot -> instance (longest expected solve first), or null
    int *queue;         // quad kernel: ticket counter of the launch (256 B, zeroed before the launch)
    const int *horizons;   // per-instance horizon 1..pb.N (nmpc_set_horizons), or null: pb.N for all
};

template <int NW, bool SQP, bool COMPACT = false>
__global__ void __launch_bounds__(64 * NW, 2) nmpc_solve_kernel(KernelArgs a)
{
    extern __shared__ double lds[];
    if ((int)blockIdx.x >= a.B) return;
    const int inst = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    nmpcw::InstanceIO io;
    io.x0 = a.x0 + (size_t)inst * 6;
    io.p = a.p + (size_t)inst * 6;
    io.W = a.W + (size_t)inst * a.w_stride;
    io.bnd = a.bnd + (size_t)inst * a.bnd_stride;
    io.traj = a.traj + (size_t)inst * (a.pb.N + 1) * 8;
    io.ctrl = a.ctrl + (size_t)inst * 6;
    io.status = a.status + inst;
    io.iters = a.iters + inst;
    io.sqp_iters = a.sqp_iters + inst;
    io.cost = a.cost + inst;
    io.cold = a.reset[inst];
    io.N = a.horizons ? a.horizons[inst] : 0;
    GpuWave<NW, SQP, COMPACT> w{lds, lds + (a.pb.N + 1) * GpuWave<NW, SQP, COMPACT>::Layout::ST + nmpcw::X_SIZE};
    nmpcw::solve_instance(w, a.pb, io);
    // instances that did not return success restart cold next time, whatever the status (NaN, QP
    // failure, SQP iteration limit): the reference re-initialises on every status != ACADOS_SUCCESS
    // (src/AcadosSolver.cpp:56-59); successful ones keep their iterate (warm start)
    if (threadIdx.x == 0) a.reset[inst] = (*io.status != 0) ? 1 : 0;
}

// ---- four instances per workgroup (nmpc_quad_solver.h): waves 0..3 are slots that do the
// element-wise work of one instance each; wave 0 also sweeps all four (one per 16-lane DPP row).
// Shares the vector-sweep stages, reductions and reciprocal with the one-wave policy.
constexpr int QUAD_THREADS = 64 * nmpcq::QI;
#ifdef NMPC_QUAD_TIMELINE
// Debug build (tools/quad_timeline.py): shader-clock ticks per phase of the interior-point iteration, summed
// over the launch per wave index: g_quad_tl[wave][segment], segment = time since the previous stamp.
__device__ unsigned long long g_quad_tl[4][16];
#endif
struct GpuQuad : GpuWave<1, false> {
#ifdef NMPC_QUAD_TIMELINE
    // accumulators of this wave in LDS (16 per wave, behind the workgroup's flags); flushed to the global
    // table once, when the workgroup ends -- per-stamp global atomics from 2048 waves on 64 addresses
    // saturate one L2 channel and slowed the launch by 1.7x
    // (state in scalar registers / LDS only: two more vector registers live through the sweeps made the
    // register-resident factor stage spill)
    int tl_off = 0;          // offset of this wave's accumulators in LDS, in 8-byte words (uniform)
    __device__ __forceinline__ unsigned long long *tl_ptr() { return (unsigned long long *)lds_ + __builtin_amdgcn_readfirstlane(tl_off); }
    __device__ __forceinline__ void stamp_init(double *area)
    {
        tl_off = __builtin_amdgcn_readfirstlane((int)(area - lds_) + 16 * wave());
        unsigned long long *tl = tl_ptr();
        if ((threadIdx.x & 63) < 15) tl[threadIdx.x & 63] = 0ull;
        if ((threadIdx.x & 63) == 15) tl[15] = (unsigned long long)__builtin_readcyclecounter();    // previous stamp
    }
    __device__ __forceinline__ void stamp(int id)
    {
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *tl = tl_ptr();
            const unsigned long long t = (unsigned long long)__builtin_readcyclecounter();
            tl[id] += t - tl[15];
            tl[15] = (unsigned long long)__builtin_readcyclecounter();
        }
    }
    __device__ __forceinline__ void tick(int id)
    {
        if ((threadIdx.x & 63) == 0) tl_ptr()[id] += 1ull;
    }
    __device__ __forceinline__ void stamp_flush()
    {
        if ((threadIdx.x & 63) < 15) atomicAdd(&g_quad_tl[wave()][threadIdx.x & 63], tl_ptr()[threadIdx.x & 63]);
    }
#else
    __device__ __forceinline__ void stamp(int) {}
    __device__ __forceinline__ void tick(int) {}
    __device__ __forceinline__ void stamp_init(double *) {}
    __device__ __forceinline__ void stamp_flush() {}
#endif
    __device__ __forceinline__ int wave() const { return (int)threadIdx.x >> 6; }
    __device__ __forceinline__ bool sweeper() const { return ((int)threadIdx.x >> 6) == 0; }   // slot 0's wave also sweeps
    // all four waves (the hardware barrier counts arriving waves, whichever branch they are in)
    __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    // next ticket of the launch-wide queue: one atomic add per wave, the same value on every lane
    __device__ __forceinline__ int next_ticket(int *queue)
    {
        int t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(t);
    }
#include "nmpc_quad_asm.inc"
};

__global__ void __launch_bounds__(QUAD_THREADS, 2) nmpc_quad_kernel(KernelArgs a)
{
    extern __shared__ double lds[];
    nmpcq::Batch bt;
    bt.B = a.B;
    bt.x0 = a.x0; bt.p = a.p; bt.W = a.W; bt.bnd = a.bnd;
    bt.w_stride = a.w_stride; bt.bnd_stride = a.bnd_stride;
    bt.traj = a.traj; bt.ctrl = a.ctrl; bt.cost = a.cost;
    bt.status = a.status; bt.iters = a.iters; bt.sqp_iters = a.sqp_iters;
    bt.reset = a.reset;
    bt.order = a.order;
    bt.queue = a.queue;
    GpuQuad w{{lds, nullptr}};
    nmpcq::solve_quad(w, a.pb, bt);
}

// ---- path front-end, un-rotation and closed-loop plant: one thread per instance
__global__ void nmpc_frontend_kernel(const double *track, int T, const double *ws, const double *pv, int B,
                                     double *x0s, double *par, double *rot, double *ws_used)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double w[6], x0[6], p[6], r;
#pragma unroll
    for (int m = 0; m < 6; ++m) { w[m] = ws[(size_t)i * 6 + m]; ws_used[(size_t)i * 6 + m] = w[m]; }
    nmpcf::frontend(track, T, w, pv[(size_t)i * 2], pv[(size_t)i * 2 + 1], x0, p, &r);
#pragma unroll
    for (int m = 0; m < 6; ++m) { x0s[(size_t)i * 6 + m] = x0[m]; par[(size_t)i * 6 + m] = p[m]; }
    rot[i] = r;
}

__global__ void nmpc_unrotate_kernel(const double *traj, const double *ws, const double *rot, int B, int N1,
                                     double *xy)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * N1) return;
    const int i = t / N1;
    double w[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) w[m] = ws[(size_t)i * 6 + m];
    double wx, wy;
    nmpcf::unrotate(w, rot[i], traj[(size_t)t * 8], traj[(size_t)t * 8 + 1], &wx, &wy);
    xy[(size_t)t * 2] = wx;
    xy[(size_t)t * 2 + 1] = wy;
}

__global__ void nmpc_plant_kernel(double *ws, const double *pv, const double *ctrl, int B, double dtp,
                                  double inv_wheelbase, double steering_ratio, int measured, double *cmd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double w[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) w[m] = ws[(size_t)i * 6 + m];
    const double dcmd = ctrl[(size_t)i * 6 + 2], tcmd = ctrl[(size_t)i * 6 + 3];
    nmpcf::plant_step(w, dcmd, tcmd, pv[(size_t)i * 2], dtp, inv_wheelbase, steering_ratio, measured);
#pragma unroll
    for (int m = 0; m < 6; ++m) ws[(size_t)i * 6 + m] = w[m];
    if (cmd) { cmd[(size_t)i * 2] = dcmd * steering_ratio; cmd[(size_t)i * 2 + 1] = tcmd; }   // what src/RosMpc.cpp:82-86 publishes
}

// actualSteeringCallback (src/RosMpc.cpp:150): currentSteeringAngle_ = msg / steeringRatio_
__global__ void nmpc_measured_steering_kernel(double *ws, const double *wheel_angle, int B, double steering_ratio)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) ws[(size_t)i * 6 + 4] = wheel_angle[i] / steering_ratio;
}

// Launch order for the next solve: instances sorted by the interior-point iterations of their
// previous solve, longest first (counting sort, one workgroup).  Workgroups are dispatched in
// blockIdx order and a CU holds 8 of them at a time, so a 4096-instance launch runs in two
// rounds per CU and ends with a ragged tail (instances take 8..17 iterations); starting the long
// ones first shortens it.  Iteration counts of consecutive calls are strongly correlated in closed
// loop (warm start, no shift); with no history (all zero) the order is the identity.  The order
// within one iteration count comes from atomics and may vary -- results do not depend on it.
constexpr int ORDER_BINS = 256;
__global__ void __launch_bounds__(1024) nmpc_order_kernel(const int *iters, int B, int *order)
{
    __shared__ int bin[ORDER_BINS];
    __shared__ int any;
    if (threadIdx.x < ORDER_BINS) bin[threadIdx.x] = 0;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const int it = iters[i];
        if (it != 0) any = 1;
        atomicAdd(&bin[it < 0 ? 0 : (it >= ORDER_BINS ? ORDER_BINS - 1 : it)], 1);
    }
    __syncthreads();
    if (!any) {     // no history: identity (uniform branch)
        for (int i = threadIdx.x; i < B; i += blockDim.x) order[i] = i;
        return;
    }
    if (threadIdx.x == 0) {     // start offsets, highest iteration count first
        int run = 0;
        for (int b = ORDER_BINS - 1; b >= 0; --b) { const int c = bin[b]; bin[b] = run; run += c; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const int it = iters[i];
        order[atomicAdd(&bin[it < 0 ? 0 : (it >= ORDER_BINS ? ORDER_BINS - 1 : it)], 1)] = i;
    }
}

// iterate [B][N+1][8] (per node X,Y,psi,v,delta,thr,u0,u1) -> the C ABI's x[B][N+1][6] and u[B][N][2]
__global__ void nmpc_split_kernel(const double *traj, int B, int N, double *xs, double *us)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N1 = (size_t)N + 1;
    if (t >= (size_t)B * N1) return;
    const size_t b = t / N1, k = t % N1;
    const double *r = traj + t * 8;
#pragma unroll
    for (int m = 0; m < 6; ++m) xs[t * 6 + m] = r[m];
    if (k < (size_t)N) { us[(b * N + k) * 2] = r[6]; us[(b * N + k) * 2 + 1] = r[7]; }
}

__global__ void nmpc_fill_rows_kernel(double *dst, int rows, int width, const double *vals8)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * width) dst[i] = vals8[i % width];
}

// cold-start flags; a reset instance also forgets its iteration history (what the launch order is
// built from): the work of a cold solve is unrelated to that of the last warm one
// warm call of nmpc_solve_static: an instance whose previous solve did not return status 0 starts cold, as
// nmpc_solve does it (the reference re-initialises on every status != ACADOS_SUCCESS, src/AcadosSolver.cpp:56-59)
__global__ void nmpc_reset_from_status_kernel(unsigned char *dst, const int *status, int B, int *iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const bool failed = status[i] != 0;
    dst[i] = failed ? 1 : 0;
    if (failed) iters[i] = 0;
}

__global__ void nmpc_set_reset_kernel(unsigned char *dst, const unsigned char *mask, int B, int *iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    if (!mask || mask[i]) { dst[i] = 1; iters[i] = 0; }
}

thread_local char g_err[256] = "";

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return false;
}

#define HIP_TRY(call)                                   \
    do {                                                \
        if (!hip_ok((call), #call)) return NMPC_EHIP;   \
    } while (0)

}  // namespace

// ------------------------------------------------------------------ host side

struct nmpc_handle {
    nmpc_config cfg;
    int capacity;
    int device;
    hipStream_t stream;
    double *d_x0, *d_p, *d_W, *d_bnd, *d_traj, *d_ctrl, *d_cost, *d_tmp;
    int *d_status, *d_iters, *d_sqp, *d_order, *d_queue;
    int *d_horizons = nullptr;    // per-instance horizons (nmpc_set_horizons), allocated on first use
    bool per_instance_N = false;
    unsigned char *d_reset, *d_mask;
    bool ordered;                 // launch longest-expected instances first (NMPC_SCHEDULE=0 turns it off)
    int hist_B = 0;               // instances [0, hist_B) may have an iteration history (0: none, the order kernel is skipped)
    KernelChoice kernel;          // NMPC_KERNEL at creation (auto / wave / quad, see use_quad_kernel)
    const double *in_x0, *in_p;   // what the kernel reads (own buffers or caller's device buffers)
    double *out_ctrl;             // where the control records go (own buffer or the caller's device buffer)
    std::vector<double> h_traj;   // nmpc_set_iterate's host image
    // trajectories out (nmpc_get / nmpc_get_view): x and u split on the device, then one DMA each into pinned
    // host memory owned by the handle; allocated at the first call that asks for trajectories
    double *d_xs = nullptr, *d_us = nullptr, *h_px = nullptr, *h_pu = nullptr;
    // inputs in (nmpc_set_x0 / nmpc_set_params): pinned staging rows [cap][6] each, so that the copy to the
    // device is a true asynchronous DMA and the call returns after one host memcpy; the event says when the
    // DMA that last read a staging buffer has finished (it must not be overwritten before).  Allocated at the
    // first upload; without them (allocation refused) the upload is a synchronous copy from the caller's memory.
    double *h_sx0 = nullptr, *h_sp = nullptr;
    hipEvent_t ev_sx0 = nullptr, ev_sp = nullptr;
    bool stage_tried = false;
    // path front-end / closed loop (allocated by nmpc_set_track)
    double *d_track = nullptr, *d_ws = nullptr, *d_pv = nullptr, *d_rot = nullptr, *d_xy = nullptr;
    double *d_wsfe = nullptr;     // the world states the last front-end pass transformed (the plant moves d_ws on)
    int n_track = 0;
};

// Problem constants of one launch from the handle-independent configuration.
static void fill_problem(const nmpc_config &c, int mode, int max_iter, double tol, nmpcw::Problem &pb)
{
    pb.N = c.N;
    pb.dt = c.dt;
    pb.inv_wheelbase = 1.0 / c.wheelbase;
    pb.ipm_max_iter = c.ipm_max_iter;
    pb.mu0 = c.ipm_mu0;
    pb.thr0 = c.ipm_thr0;
    pb.tau = c.ipm_tau;
    pb.tol_mu = c.ipm_tol_mu;
    pb.tol_lin = c.ipm_tol_lin;
    pb.tol_comp = c.ipm_tol_comp;
    pb.stall_iter = c.ipm_stall_iter;
    pb.stall_lin = c.ipm_stall_lin;
    pb.sqp_max_iter = (mode == NMPC_MODE_RTI) ? 1 : max_iter;
    pb.sqp_tol = tol;
    pb.polish = c.polish;
    pb.polish_mu = c.polish_mu;
    pb.polish_tries = c.polish_tries;
    pb.polish_tol = c.polish_tol;
    pb.warm_tries = c.warm_tries;
    pb.warm_gate = c.warm_gate;
    pb.warm_eps = c.warm_eps;
}

// Which solve kernel.  Two implement SQP-RTI up to N = 63: one instance per workgroup
// (nmpc_wave_solver.h) and four slots per workgroup fed from a ticket queue (nmpc_quad_solver.h).
// Measured (profiles/r2/quad_vs_wave_by_horizon_after_symmetrisation.txt): the four-slot kernel is 1-8 % faster where two
// of its workgroups fit into a CU's LDS (N <= 43) and the sweeps are long enough to pay for its four
// workgroup barriers per iteration (N >= 32), once the batch fills the chip several times over; it is
// slower for short horizons, by 25 % where only one workgroup fits, and has the longer single-solve
// latency.  NMPC_KERNEL=quad / wave forces one of them (read when a handle is created / per call of
// nmpc_solve_static); everything else -- SQP mode, N > 63, per-instance horizons -- runs on the
// one-instance kernel.
static KernelChoice kernel_choice()
{
    const char *e = getenv("NMPC_KERNEL");
    if (e && strcmp(e, "quad") == 0) return KERNEL_QUAD;
    if (e && strcmp(e, "wave") == 0) return KERNEL_WAVE;
    return KERNEL_AUTO;
}
// Compute units and LDS per compute unit of a device, asked once per device (MI355X: 256 and 160 KiB; other
// parts or partition modes differ, and the grid of the four-slot kernel and its LDS test follow them).
struct DeviceShape { int cus, lds_per_cu; };
static DeviceShape device_shape(int device)
{
    static std::mutex mu;
    static DeviceShape cache[64];
    static bool have[64];
    std::lock_guard<std::mutex> lk(mu);
    const int d = (device >= 0 && device < 64) ? device : 0;
    if (!have[d]) {
        DeviceShape s{256, 160 * 1024};
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, d) == hipSuccess) {
            if (pr.multiProcessorCount > 0) s.cus = pr.multiProcessorCount;
            if (pr.maxSharedMemoryPerMultiProcessor > 0) s.lds_per_cu = (int)pr.maxSharedMemoryPerMultiProcessor;
        }
        cache[d] = s;
        have[d] = true;
    }
    return cache[d];
}

static bool use_quad_kernel(KernelChoice c, int mode, int N, int B, bool polish, int lds_per_cu = 160 * 1024)
{
    if (mode != NMPC_MODE_RTI || nmpcw::waves_per_instance(N) != 1 || c == KERNEL_WAVE) return false;
    if (c == KERNEL_QUAD) return true;
    // the four-slot kernel runs the interior-point iteration to the end (no active-set polish): with the
    // polish on, the one-instance kernel needs a third fewer factorisations and is the faster one
    if (polish) return false;
    const size_t qbytes = (size_t)nmpcq::quad_lds_doubles(N) * sizeof(double);
    return B >= 8192 && N >= 32 && 2 * qbytes <= (size_t)lds_per_cu;
}

// Order kernel (optional: `hist` = iteration counts of the previous solve, `order` = its output) and
// the solve kernel of one batch on `stream`.  Shared by nmpc_solve and nmpc_solve_static.
static int launch_solve(KernelArgs &a, int mode, hipStream_t stream, const int *hist, int *order, KernelChoice choice)
{
    const int B = a.B, N = a.pb.N;
    a.order = nullptr;
    // (beyond ~8 rounds per CU the tail is a small part of the launch and the single-workgroup sort is not free)
    if (hist && order && B > 1 && B <= 16384) {
        hipLaunchKernelGGL(nmpc_order_kernel, dim3(1), dim3(1024), 0, stream, hist, B, order);
        HIP_TRY(hipGetLastError());
        a.order = order;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    const DeviceShape shape = device_shape(dev);
    if (use_quad_kernel(choice, mode, N, B, a.pb.polish != 0, shape.lds_per_cu)) {
        // SQP-RTI up to N = 63: workgroups of four waves, each wave a slot that draws instances from a
        // launch-wide ticket queue (nmpc_quad_solver.h); no more workgroups than can be resident at once
        // are needed -- later ones would find the queue empty
        if (!a.queue) return NMPC_EINVAL;
        HIP_TRY(hipMemsetAsync(a.queue, 0, 256, stream));
        const size_t qbytes = (size_t)nmpcq::quad_lds_doubles(N) * sizeof(double);
        const int fit = (int)((size_t)shape.lds_per_cu / qbytes);
        const int per_cu = fit < 2 ? fit : 2;       // four waves of up to 256 VGPRs each: two workgroups per CU
        int wgs = (B + nmpcq::QI - 1) / nmpcq::QI;
        const int resident = shape.cus * (per_cu < 1 ? 1 : per_cu);
        if (wgs > resident) wgs = resident;
        hipLaunchKernelGGL(nmpc_quad_kernel, dim3(wgs), dim3(QUAD_THREADS), qbytes, stream, a);
        HIP_TRY(hipGetLastError());
        return NMPC_OK;
    }
    const size_t lds_bytes = (size_t)nmpcw::lds_doubles(N) * sizeof(double);
    // one workgroup per instance: one wave (lane = shooting node) up to N = 63, two waves beyond
    const bool one = nmpcw::waves_per_instance(N) == 1;
    // (two waves: the 48-double stage where it lets one more instance reside per CU, nmpcw::compact_layout)
    const bool compact = nmpcw::compact_layout(N);
    if (mode == NMPC_MODE_RTI) {
        if (one && compact) hipLaunchKernelGGL((nmpc_solve_kernel<1, false, true>), dim3(B), dim3(64), lds_bytes, stream, a);
        else if (one) hipLaunchKernelGGL((nmpc_solve_kernel<1, false>), dim3(B), dim3(64), lds_bytes, stream, a);
        else if (compact) hipLaunchKernelGGL((nmpc_solve_kernel<2, false, true>), dim3(B), dim3(128), lds_bytes, stream, a);
        else hipLaunchKernelGGL((nmpc_solve_kernel<2, false>), dim3(B), dim3(128), lds_bytes, stream, a);
    } else {
        if (one && compact) hipLaunchKernelGGL((nmpc_solve_kernel<1, true, true>), dim3(B), dim3(64), lds_bytes, stream, a);
        else if (one) hipLaunchKernelGGL((nmpc_solve_kernel<1, true>), dim3(B), dim3(64), lds_bytes, stream, a);
        else if (compact) hipLaunchKernelGGL((nmpc_solve_kernel<2, true, true>), dim3(B), dim3(128), lds_bytes, stream, a);
        else hipLaunchKernelGGL((nmpc_solve_kernel<2, true>), dim3(B), dim3(128), lds_bytes, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return NMPC_OK;
}

// Horizons whose LDS image exceeds the 48 KiB a kernel gets by default need the limit raised once.
static int raise_lds_limit(int N)
{
    if (nmpcw::waves_per_instance(N) == 1) {
        const int qbytes = nmpcq::quad_lds_doubles(N) * (int)sizeof(double);
        if (qbytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)nmpc_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, qbytes));
    }
    const int lds_bytes = nmpcw::lds_doubles(N) * (int)sizeof(double);
    if (lds_bytes <= 48 * 1024) return NMPC_OK;
    const bool one = nmpcw::waves_per_instance(N) == 1, compact = nmpcw::compact_layout(N);
    const void *fns[2] = {one ? (compact ? (const void *)nmpc_solve_kernel<1, false, true> : (const void *)nmpc_solve_kernel<1, false>)
                              : (compact ? (const void *)nmpc_solve_kernel<2, false, true> : (const void *)nmpc_solve_kernel<2, false>),
                          one ? (compact ? (const void *)nmpc_solve_kernel<1, true, true> : (const void *)nmpc_solve_kernel<1, true>)
                              : (compact ? (const void *)nmpc_solve_kernel<2, true, true> : (const void *)nmpc_solve_kernel<2, true>)};
    for (const void *fn : fns)
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    return NMPC_OK;
}

// ---- one entry point that allocates nothing (include/nmpc.h: nmpc_solve_static).  Everything the
// solve needs beyond the caller's inputs and outputs -- the launch order, the iteration history it
// is built from, the cold-start flags, per-instance SQP counters and costs, one row of default
// weights and bounds -- lives in a file-scope device array; each GPU holds its own copy, whose
// address is looked up once per device.
struct StaticScratch {
    double cost[NMPC_STATIC_MAX_BATCH];
    double row[16];                         // W[7] | pad | bnd[8] of the configuration
    int order[NMPC_STATIC_MAX_BATCH];
    int hist[NMPC_STATIC_MAX_BATCH];
    int sqp[NMPC_STATIC_MAX_BATCH];
    int queue[64];                          // ticket counter of the launch (256 B)
    unsigned char reset[NMPC_STATIC_MAX_BATCH];
};
static __device__ __align__(256) unsigned char scratch[sizeof(StaticScratch)];

struct RowArg { double v[16]; };
static __global__ void nmpc_static_row_kernel(double *dst, RowArg r)
{
    if (threadIdx.x < 16) dst[threadIdx.x] = r.v[threadIdx.x];
}

constexpr int STATIC_MAX_DEVICES = 64;
static StaticScratch *g_static_ptr[STATIC_MAX_DEVICES];
static bool g_static_lds_raised[STATIC_MAX_DEVICES][128];
static std::mutex g_static_mu;

// device address of `scratch` on the CURRENT device (must be `device`), looked up once per device:
// the array's name is only a host-side stand-in, never a device address
static int static_scratch(int device, StaticScratch **out)
{
    std::lock_guard<std::mutex> lk(g_static_mu);
    if (!g_static_ptr[device]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_static_ptr[device] = static_cast<StaticScratch *>(p);
    }
    *out = g_static_ptr[device];
    return NMPC_OK;
}

extern "C" {

const char *nmpc_last_error(void) { return g_err; }

int nmpc_default_config(nmpc_config *c, int N, double dt)
{
    if (!c) return NMPC_EINVAL;
    static const double W[7] = {5, 5, 10, 0.01, 0.1, 0.5, 0.1};
    memset(c, 0, sizeof(*c));
    c->N = N;
    c->dt = dt;
    c->wheelbase = 2.65;
    memcpy(c->W, W, sizeof(W));
    c->lbu[0] = -80.0; c->ubu[0] = 80.0;
    c->lbu[1] = -0.33; c->ubu[1] = 0.33;
    c->lbx[0] = -0.57; c->ubx[0] = 0.57;
    c->lbx[1] = -1.0;  c->ubx[1] = 1.0;
    c->ipm_max_iter = 50;
    c->ipm_mu0 = 100.0;
    c->ipm_thr0 = 0.5;
    c->ipm_tau = 0.995;
    c->ipm_tol_mu = 1e-10;
    c->ipm_tol_lin = 1e-10;
    c->ipm_tol_comp = 1e-11;
    c->ipm_stall_iter = 20;
    c->ipm_stall_lin = 0.3;
    c->polish = 1;
    c->polish_mu = 1e-3;
    c->polish_tries = 2;
    c->polish_tol = 1e-10;
    c->warm_tries = 1;
    c->warm_gate = 2;
    c->warm_eps = 1e-9;
    return NMPC_OK;
}

int nmpc_config_from_file(const char *path, nmpc_config *cfg, nmpc_run_params *run)
try {
    if (!path || !cfg) return NMPC_EINVAL;
    FILE *f = fopen(path, "r");
    if (!f) {
        snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: cannot open %s", path);
        return NMPC_EINVAL;
    }
    nmpcy::Table t;
    std::string err;
    const bool parsed = nmpcy::parse(f, t, err);
    fclose(f);
    if (!parsed) {
        snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: %s", path, err.c_str());
        return NMPC_EINVAL;
    }
    std::string pre;
    if (!nmpcy::find_prefix(t, "mpc_N", &pre)) {
        snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: missing key mpc_N", path);
        return NMPC_EINVAL;
    }
    // required keys: the ones the reference's code generation indexes (mpcPathTracking.py:9,91-92,105-108,126)
    auto numbers = [&](const char *key, int n, double *out, bool required) -> int {
        nmpcy::Table::const_iterator it = t.find(pre + key);
        if (it == t.end()) {
            if (!required) return 1;
            snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: missing key %s", path, key);
            return -1;
        }
        if ((int)it->second.size() != n) {
            snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: %s needs %d value(s)", path, key, n);
            return -1;
        }
        for (int i = 0; i < n; ++i)
            if (!nmpcy::to_double(it->second[i], &out[i])) {
                snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: %s: '%s' is not a number", path, key,
                         it->second[i].c_str());
                return -1;
            }
        return 0;
    };
    double nn = 0.0, dt = 0.0, wb = 0.0, r[4][2];
    if (numbers("mpc_N", 1, &nn, true) < 0 || numbers("mpc_dt", 1, &dt, true) < 0 ||
        numbers("wheelbase", 1, &wb, true) < 0 || numbers("steering_angle_dot_range", 2, r[0], true) < 0 ||
        numbers("throttle_dot_range", 2, r[1], true) < 0 || numbers("steering_angle_range", 2, r[2], true) < 0 ||
        numbers("throttle_range", 2, r[3], true) < 0)
        return NMPC_EINVAL;
    if (nn != (double)(int)nn || nn < 1 || nn > 127 || !(dt > 0.0) || !(wb > 0.0)) {
        snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: mpc_N in 1..127, mpc_dt > 0, wheelbase > 0", path);
        return NMPC_EINVAL;
    }
    for (int i = 0; i < 4; ++i)
        if (!(r[i][0] < r[i][1])) {
            snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: a range needs lower < upper", path);
            return NMPC_EINVAL;
        }
    nmpcy::Table::const_iterator st = t.find(pre + "nlp_solver_type");
    int mode;
    if (st == t.end() || st->second.size() != 1) {
        snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: missing key nlp_solver_type", path);
        return NMPC_EINVAL;
    } else if (st->second[0] == "SQP_RTI") {
        mode = NMPC_MODE_RTI;
    } else if (st->second[0] == "SQP") {
        mode = NMPC_MODE_SQP;
    } else {
        snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: nlp_solver_type '%s' (SQP or SQP_RTI)", path,
                 st->second[0].c_str());
        return NMPC_EINVAL;
    }
    nmpc_run_params rp;
    rp.mode = mode;
    rp.loop_hz = 30.0;              // config/mpc_params.yaml:9, src/main.cpp:14
    rp.path_tracking_vel = 5.0;     // config/mpc_params.yaml:44
    rp.steering_ratio = 1.0;        // default of src/RosMpc.cpp:34
    if (numbers("loop_Hz", 1, &rp.loop_hz, false) < 0 || numbers("path_tracking_vel", 1, &rp.path_tracking_vel, false) < 0 ||
        numbers("steering_ratio", 1, &rp.steering_ratio, false) < 0)
        return NMPC_EINVAL;
    if (!(rp.loop_hz > 0.0) || rp.steering_ratio == 0.0) {
        snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: %s: loop_Hz > 0 and steering_ratio != 0", path);
        return NMPC_EINVAL;
    }
    nmpc_default_config(cfg, (int)nn, dt);
    cfg->wheelbase = wb;
    cfg->lbu[0] = r[0][0]; cfg->ubu[0] = r[0][1];
    cfg->lbu[1] = r[1][0]; cfg->ubu[1] = r[1][1];
    cfg->lbx[0] = r[2][0]; cfg->ubx[0] = r[2][1];
    cfg->lbx[1] = r[3][0]; cfg->ubx[1] = r[3][1];
    if (run) *run = rp;
    return NMPC_OK;
} catch (const std::bad_alloc &) {   // host containers: no exception may cross the C ABI
    snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: out of host memory");
    return NMPC_ENOMEM;
} catch (...) {
    snprintf(g_err, sizeof(g_err), "nmpc_config_from_file: unexpected exception");
    return NMPC_EINVAL;
}

// W <= 0 makes the 2x2 input Hessian singular (every solve would return status 1), lb >= ub leaves
// no interior for the interior-point iteration: rejected where they enter
static bool valid_weights(const double *W, int rows)
{
    for (int i = 0; i < rows * 7; ++i)
        if (!(W[i] > 0.0) || !(W[i] < 1e300)) {
            snprintf(g_err, sizeof(g_err), "weights: W[%d][%d] = %g, every weight must be positive and finite", i / 7, i % 7, W[i]);
            return false;
        }
    return true;
}

static bool valid_bounds(const double *bnd, int rows)
{
    for (int i = 0; i < rows; ++i)
        for (int q = 0; q < 4; ++q)
            if (!(bnd[i * 8 + q] < bnd[i * 8 + 4 + q])) {
                snprintf(g_err, sizeof(g_err), "bounds: row %d, bound %d: lower %g is not below upper %g", i, q,
                         bnd[i * 8 + q], bnd[i * 8 + 4 + q]);
                return false;
            }
    return true;
}

static bool valid_config(const nmpc_config *c)
{
    const double b[8] = {c->lbu[0], c->lbu[1], c->lbx[0], c->lbx[1], c->ubu[0], c->ubu[1], c->ubx[0], c->ubx[1]};
    if (c->N < 1 || c->N > 127 || !(c->dt > 0.0) || !(c->wheelbase > 0.0)) {
        snprintf(g_err, sizeof(g_err), "config: N in 1..127, dt > 0, wheelbase > 0");
        return false;
    }
    return valid_weights(c->W, 1) && valid_bounds(b, 1);
}

static int fill_defaults(nmpc_handle *h, bool weights, bool bounds)
{
    const int cap = h->capacity;
    double v[16];
    if (weights) {
        HIP_TRY(hipMemcpyAsync(h->d_tmp, h->cfg.W, sizeof(double) * 7, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(nmpc_fill_rows_kernel, dim3((cap * 7 + 255) / 256), dim3(256), 0, h->stream,
                           h->d_W, cap, 7, h->d_tmp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (bounds) {
        v[0] = h->cfg.lbu[0]; v[1] = h->cfg.lbu[1]; v[2] = h->cfg.lbx[0]; v[3] = h->cfg.lbx[1];
        v[4] = h->cfg.ubu[0]; v[5] = h->cfg.ubu[1]; v[6] = h->cfg.ubx[0]; v[7] = h->cfg.ubx[1];
        HIP_TRY(hipMemcpyAsync(h->d_tmp, v, sizeof(double) * 8, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(nmpc_fill_rows_kernel, dim3((cap * 8 + 255) / 256), dim3(256), 0, h->stream,
                           h->d_bnd, cap, 8, h->d_tmp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return NMPC_OK;
}

int nmpc_create(const nmpc_config *cfg, int capacity, int device, nmpc_handle **out)
{
    if (!cfg || !out || capacity <= 0) return NMPC_EINVAL;
    if (!valid_config(cfg)) return NMPC_EINVAL;
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(device));
    nmpc_handle *h = new (std::nothrow) nmpc_handle();
    if (!h) return NMPC_ENOMEM;
    h->cfg = *cfg;
    h->capacity = capacity;
    h->device = device;
    const size_t cap = (size_t)capacity, N1 = (size_t)cfg->N + 1;
    struct { void **p; size_t bytes; } allocs[] = {
        {(void **)&h->d_x0, cap * 6 * sizeof(double)},   {(void **)&h->d_p, cap * 6 * sizeof(double)},
        {(void **)&h->d_W, cap * 7 * sizeof(double)},    {(void **)&h->d_bnd, cap * 8 * sizeof(double)},
        {(void **)&h->d_traj, cap * N1 * 8 * sizeof(double)}, {(void **)&h->d_ctrl, cap * 6 * sizeof(double)},
        {(void **)&h->d_cost, cap * sizeof(double)},     {(void **)&h->d_tmp, 16 * sizeof(double)},
        {(void **)&h->d_status, cap * sizeof(int)},      {(void **)&h->d_iters, cap * sizeof(int)},
        {(void **)&h->d_sqp, cap * sizeof(int)},         {(void **)&h->d_reset, cap},
        {(void **)&h->d_mask, cap},                      {(void **)&h->d_order, cap * sizeof(int)},
        {(void **)&h->d_queue, 256},
    };
    const char *sched = getenv("NMPC_SCHEDULE");
    h->ordered = !(sched && sched[0] == '0');
    h->kernel = kernel_choice();
    for (auto &a : allocs) *a.p = nullptr;
    bool ok = hip_ok(hipStreamCreate(&h->stream), "hipStreamCreate");
    for (auto &a : allocs) {
        if (!ok) break;
        ok = hip_ok(hipMalloc(a.p, a.bytes), "hipMalloc");
        if (ok) ok = hip_ok(hipMemsetAsync(*a.p, 0, a.bytes, h->stream), "hipMemsetAsync");
    }
    if (ok) ok = hip_ok(hipMemsetAsync(h->d_reset, 1, cap, h->stream), "hipMemsetAsync");
    if (ok) ok = fill_defaults(h, true, true) == NMPC_OK;
    if (ok) ok = raise_lds_limit(cfg->N) == NMPC_OK;
    if (!ok) {
        for (auto &a : allocs) if (*a.p) (void)hipFree(*a.p);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        delete h;
        return NMPC_EHIP;
    }
    h->in_x0 = h->d_x0;
    h->in_p = h->d_p;
    h->out_ctrl = h->d_ctrl;
    *out = h;
    return NMPC_OK;
}

int nmpc_destroy(nmpc_handle *h)
{
    if (!h) return NMPC_EINVAL;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->d_x0, h->d_p, h->d_W, h->d_bnd, h->d_traj, h->d_ctrl, h->d_cost, h->d_tmp,
                    h->d_status, h->d_iters, h->d_sqp, h->d_order, h->d_queue, h->d_reset, h->d_mask,
                    h->d_track, h->d_ws, h->d_pv, h->d_rot, h->d_xy, h->d_wsfe, h->d_horizons};
    int rc = NMPC_OK;
    for (void *p : ptrs)
        if (p && !hip_ok(hipFree(p), "hipFree")) rc = NMPC_EHIP;
    for (void *p : {(void *)h->d_xs, (void *)h->d_us})
        if (p && !hip_ok(hipFree(p), "hipFree")) rc = NMPC_EHIP;
    for (void *p : {(void *)h->h_px, (void *)h->h_pu, (void *)h->h_sx0, (void *)h->h_sp})
        if (p && !hip_ok(hipHostFree(p), "hipHostFree")) rc = NMPC_EHIP;
    for (hipEvent_t e : {h->ev_sx0, h->ev_sp})
        if (e && !hip_ok(hipEventDestroy(e), "hipEventDestroy")) rc = NMPC_EHIP;
    if (!hip_ok(hipStreamDestroy(h->stream), "hipStreamDestroy")) rc = NMPC_EHIP;
    delete h;
    return rc;
}

static int upload(nmpc_handle *h, double *dst, const double *src, int B, int width, double *nmpc_handle::*stage = nullptr,
                  hipEvent_t nmpc_handle::*ev = nullptr)
{
    if (!h || !src || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = sizeof(double) * (size_t)B * width;
    if (stage && !h->stage_tried) {
        // once per handle, all or nothing (a refusal is not an error: the synchronous path below remains)
        h->stage_tried = true;
        const size_t cap_bytes = sizeof(double) * (size_t)h->capacity * 6;
        double *a = nullptr, *b = nullptr;
        hipEvent_t ea = nullptr, eb = nullptr;
        if (hipHostMalloc((void **)&a, cap_bytes, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void **)&b, cap_bytes, hipHostMallocDefault) == hipSuccess &&
            hipEventCreateWithFlags(&ea, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&eb, hipEventDisableTiming) == hipSuccess) {
            h->h_sx0 = a; h->h_sp = b; h->ev_sx0 = ea; h->ev_sp = eb;
        } else {
            (void)hipGetLastError();
            if (a) (void)hipHostFree(a);
            if (b) (void)hipHostFree(b);
            if (ea) (void)hipEventDestroy(ea);
            if (eb) (void)hipEventDestroy(eb);
        }
    }
    if (stage && h->*stage) {
        HIP_TRY(hipEventSynchronize(h->*ev));          // the previous DMA out of this staging buffer (long done, as a rule)
        memcpy(h->*stage, src, bytes);                  // the caller's buffer may be reused as soon as we return
        HIP_TRY(hipMemcpyAsync(dst, h->*stage, bytes, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipEventRecord(h->*ev, h->stream));
        return NMPC_OK;
    }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    // the caller's buffer may be reused as soon as we return
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NMPC_OK;
}

int nmpc_set_x0(nmpc_handle *h, const double *x0, int B)
{
    const int rc = upload(h, h ? h->d_x0 : nullptr, x0, B, 6, &nmpc_handle::h_sx0, &nmpc_handle::ev_sx0);
    if (rc == NMPC_OK) h->in_x0 = h->d_x0;
    return rc;
}

int nmpc_set_params(nmpc_handle *h, const double *p, int B)
{
    const int rc = upload(h, h ? h->d_p : nullptr, p, B, 6, &nmpc_handle::h_sp, &nmpc_handle::ev_sp);
    if (rc == NMPC_OK) h->in_p = h->d_p;
    return rc;
}

int nmpc_set_inputs_device(nmpc_handle *h, const double *d_x0, const double *d_p, int B)
{
    if (!h || !d_x0 || !d_p || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    h->in_x0 = d_x0;
    h->in_p = d_p;
    return NMPC_OK;
}

int nmpc_set_controls_device(nmpc_handle *h, double *d_ctrl)
{
    if (!h) return NMPC_EINVAL;
    h->out_ctrl = d_ctrl ? d_ctrl : h->d_ctrl;
    return NMPC_OK;
}

int nmpc_set_weights(nmpc_handle *h, const double *W, int B)
{
    if (!h) return NMPC_EINVAL;
    if (!W) { HIP_TRY(hipSetDevice(h->device)); return fill_defaults(h, true, false); }
    if (B <= 0 || B > h->capacity || !valid_weights(W, B)) return NMPC_EINVAL;
    return upload(h, h->d_W, W, B, 7);
}

int nmpc_set_bounds(nmpc_handle *h, const double *bnd, int B)
{
    if (!h) return NMPC_EINVAL;
    if (!bnd) { HIP_TRY(hipSetDevice(h->device)); return fill_defaults(h, false, true); }
    if (B <= 0 || B > h->capacity || !valid_bounds(bnd, B)) return NMPC_EINVAL;
    return upload(h, h->d_bnd, bnd, B, 8);
}

int nmpc_set_horizons(nmpc_handle *h, const int32_t *N, int B)
{
    if (!h) return NMPC_EINVAL;
    if (!N) {
        // every instance uses cfg.N again -- also in the table, so that a later call naming fewer instances
        // does not bring stale horizons back for the others
        h->per_instance_N = false;
        if (h->d_horizons) {
            HIP_TRY(hipSetDevice(h->device));
            std::vector<int> all;
            try { all.assign((size_t)h->capacity, h->cfg.N); } catch (...) { return NMPC_ENOMEM; }
            HIP_TRY(hipMemcpyAsync(h->d_horizons, all.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        return NMPC_OK;
    }
    if (B <= 0 || B > h->capacity) return NMPC_EINVAL;
    for (int i = 0; i < B; ++i)
        if (N[i] < 1 || N[i] > h->cfg.N) {
            snprintf(g_err, sizeof(g_err), "nmpc_set_horizons: N[%d] = %d, must be in 1..%d (the handle's horizon)", i, N[i], h->cfg.N);
            return NMPC_EINVAL;
        }
    HIP_TRY(hipSetDevice(h->device));
    if (!h->d_horizons) {
        HIP_TRY(hipMalloc((void **)&h->d_horizons, sizeof(int) * (size_t)h->capacity));
        // instances not named keep the handle's horizon
        std::vector<int> all;
        try { all.assign((size_t)h->capacity, h->cfg.N); } catch (...) { return NMPC_ENOMEM; }
        HIP_TRY(hipMemcpyAsync(h->d_horizons, all.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    HIP_TRY(hipMemcpyAsync(h->d_horizons, N, sizeof(int) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->per_instance_N = true;
    return NMPC_OK;
}

int nmpc_reset(nmpc_handle *h, const uint8_t *mask, int B)
{
    if (!h || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    if (mask) HIP_TRY(hipMemcpyAsync(h->d_mask, mask, (size_t)B, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(nmpc_set_reset_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->d_reset,
                       mask ? h->d_mask : nullptr, B, h->d_iters);
    HIP_TRY(hipGetLastError());
    if (mask) HIP_TRY(hipStreamSynchronize(h->stream));
    else if (B >= h->hist_B) h->hist_B = 0;      // everything that had a history has lost it
    return NMPC_OK;
}

int nmpc_set_iterate(nmpc_handle *h, const double *x, const double *u, int B)
try {
    if (!h || !x || !u || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    const size_t N = (size_t)h->cfg.N, N1 = N + 1;
    h->h_traj.resize((size_t)B * N1 * 8);
    double *t = h->h_traj.data();
    for (size_t b = 0; b < (size_t)B; ++b)
        for (size_t k = 0; k < N1; ++k) {
            double *r = t + (b * N1 + k) * 8;
            memcpy(r, x + (b * N1 + k) * 6, sizeof(double) * 6);
            r[6] = k < N ? u[(b * N + k) * 2] : 0.0;
            r[7] = k < N ? u[(b * N + k) * 2 + 1] : 0.0;
        }
    HIP_TRY(hipMemcpyAsync(h->d_traj, t, sizeof(double) * h->h_traj.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->d_reset, 0, (size_t)B, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NMPC_OK;
} catch (const std::bad_alloc &) {   // host containers: no exception may cross the C ABI
    snprintf(g_err, sizeof(g_err), "nmpc_set_iterate: out of host memory");
    return NMPC_ENOMEM;
} catch (...) {
    snprintf(g_err, sizeof(g_err), "nmpc_set_iterate: unexpected exception");
    return NMPC_EINVAL;
}

int nmpc_solve(nmpc_handle *h, int B, int mode, int max_iter, double tol)
{
    if (!h || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    if (mode != NMPC_MODE_RTI && mode != NMPC_MODE_SQP) return NMPC_EINVAL;
    if (mode == NMPC_MODE_SQP && max_iter < 1) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    KernelArgs a;
    fill_problem(h->cfg, mode, max_iter, tol, a.pb);
    a.B = B;
    a.x0 = h->in_x0; a.p = h->in_p; a.W = h->d_W; a.bnd = h->d_bnd;
    a.w_stride = 7; a.bnd_stride = 8;
    a.traj = h->d_traj; a.ctrl = h->out_ctrl; a.cost = h->d_cost;
    a.status = h->d_status; a.iters = h->d_iters; a.sqp_iters = h->d_sqp;
    a.reset = h->d_reset;
    a.queue = h->d_queue;
    a.horizons = h->per_instance_N ? h->d_horizons : nullptr;
    // (the four-slot kernel sweeps four instances in lock step: one horizon per launch)
    const int *hist = (h->ordered && h->hist_B > 0) ? h->d_iters : nullptr;   // nothing to order by after a full reset
    if (B > h->hist_B) h->hist_B = B;
    return launch_solve(a, mode, h->stream, hist, h->d_order, h->per_instance_N ? KERNEL_WAVE : h->kernel);
}

#ifdef NMPC_WAVE_TIMELINE
template <int NW, bool SQP, bool COMPACT>
__device__ __forceinline__ void GpuWave<NW, SQP, COMPACT>::stamp_flush()
{
    __syncthreads();
    if (threadIdx.x < 23) atomicAdd(&g_wave_tl[threadIdx.x], tl_ptr()[threadIdx.x]);
}
extern "C" int nmpc_debug_wave_timeline(unsigned long long *out, int clear)
{
    if (out) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_tl), sizeof(unsigned long long) * 24));
    if (clear) {
        unsigned long long zero[24] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_tl), zero, sizeof(zero)));
    }
    return NMPC_OK;
}
#endif

#ifdef NMPC_QUAD_TIMELINE
extern "C" int nmpc_debug_quad_timeline(unsigned long long *out, int clear)
{
    if (out) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_tl), sizeof(unsigned long long) * 64));
    if (clear) {
        unsigned long long zero[64] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_quad_tl), zero, sizeof(zero)));
    }
    return NMPC_OK;
}
#endif

int nmpc_solve_static(const nmpc_config *cfg, int device, void *stream, int B, int cold, const double *d_x0,
                      const double *d_p, const double *d_W, const double *d_bnd, double *d_traj, double *d_ctrl,
                      int32_t *d_status, int32_t *d_ipm_iters)
{
    if (!cfg || !d_x0 || !d_p || !d_traj || !d_ctrl || !d_status || !d_ipm_iters || B <= 0) {
        snprintf(g_err, sizeof(g_err), "nmpc_solve_static: null argument or B <= 0");
        return NMPC_EINVAL;
    }
    if (B > NMPC_STATIC_MAX_BATCH) {
        snprintf(g_err, sizeof(g_err), "nmpc_solve_static: B = %d exceeds NMPC_STATIC_MAX_BATCH = %d (the static "
                 "device workspace is sized for that many instances; use nmpc_create/nmpc_solve beyond it)", B,
                 NMPC_STATIC_MAX_BATCH);
        return NMPC_EINVAL;
    }
    if (!valid_config(cfg)) return NMPC_EINVAL;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || device >= STATIC_MAX_DEVICES) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(device));
    StaticScratch *sc = nullptr;
    int rc = static_scratch(device, &sc);
    if (rc != NMPC_OK) return rc;
    {
        std::lock_guard<std::mutex> lk(g_static_mu);
        if (!g_static_lds_raised[device][cfg->N]) {
            rc = raise_lds_limit(cfg->N);
            if (rc != NMPC_OK) return rc;
            g_static_lds_raised[device][cfg->N] = true;
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the array keeps the previous call's contents: flags are rewritten for this batch -- all cold, or, on a
    // warm call, cold where the caller's d_status says the previous solve of the instance failed (what the
    // kernels of nmpc_solve leave in the handle's flags) -- and a cold batch has no history worth ordering
    // by (zero = identity order)
    if (cold) {
        HIP_TRY(hipMemsetAsync(sc->reset, 1, (size_t)B, st));
    } else {
        HIP_TRY(hipMemcpyAsync(sc->hist, d_ipm_iters, sizeof(int) * (size_t)B, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(nmpc_reset_from_status_kernel, dim3((B + 255) / 256), dim3(256), 0, st, sc->reset, d_status, B, sc->hist);
        HIP_TRY(hipGetLastError());
    }
    KernelArgs a;
    fill_problem(*cfg, NMPC_MODE_RTI, 1, 0.0, a.pb);
    a.B = B;
    a.x0 = d_x0; a.p = d_p;
    a.W = d_W; a.w_stride = 7;
    a.bnd = d_bnd; a.bnd_stride = 8;
    if (!d_W || !d_bnd) {     // one shared row of the configuration's values
        RowArg r;
        memset(&r, 0, sizeof(r));
        memcpy(r.v, cfg->W, sizeof(double) * 7);
        r.v[8] = cfg->lbu[0]; r.v[9] = cfg->lbu[1]; r.v[10] = cfg->lbx[0]; r.v[11] = cfg->lbx[1];
        r.v[12] = cfg->ubu[0]; r.v[13] = cfg->ubu[1]; r.v[14] = cfg->ubx[0]; r.v[15] = cfg->ubx[1];
        hipLaunchKernelGGL(nmpc_static_row_kernel, dim3(1), dim3(64), 0, st, sc->row, r);
        HIP_TRY(hipGetLastError());
        if (!d_W) { a.W = sc->row; a.w_stride = 0; }
        if (!d_bnd) { a.bnd = sc->row + 8; a.bnd_stride = 0; }
    }
    a.traj = d_traj; a.ctrl = d_ctrl; a.cost = sc->cost;
    a.status = d_status; a.iters = d_ipm_iters; a.sqp_iters = sc->sqp;
    a.reset = sc->reset;
    a.queue = sc->queue;
    a.horizons = nullptr;
    return launch_solve(a, NMPC_MODE_RTI, st, cold ? nullptr : sc->hist, sc->order, kernel_choice());
}

int nmpc_shard_range(int64_t total, int rank, int world, int64_t *begin, int64_t *count)
{
    if (!begin || !count || world <= 0 || rank < 0 || rank >= world || total < 0) return NMPC_EINVAL;
    const int64_t base = total / world, rem = total % world;
    *begin = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1 : 0);
    return NMPC_OK;
}

int nmpc_sync(nmpc_handle *h)
{
    if (!h) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NMPC_OK;
}

// x / u of instances 0..B-1 into the handle's pinned host buffers (enqueued; the caller synchronises)
static int stage_trajectories(nmpc_handle *h, int B)
{
    const size_t cap = (size_t)h->capacity, N = (size_t)h->cfg.N, N1 = N + 1;
    if (!h->d_xs) {
        // all or nothing, once per handle
        double *dx = nullptr, *du = nullptr, *px = nullptr, *pu = nullptr;
        bool ok = hip_ok(hipMalloc((void **)&dx, sizeof(double) * cap * N1 * 6), "hipMalloc") &&
                  hip_ok(hipMalloc((void **)&du, sizeof(double) * cap * N * 2), "hipMalloc") &&
                  hip_ok(hipHostMalloc((void **)&px, sizeof(double) * cap * N1 * 6, hipHostMallocDefault), "hipHostMalloc") &&
                  hip_ok(hipHostMalloc((void **)&pu, sizeof(double) * cap * N * 2, hipHostMallocDefault), "hipHostMalloc");
        if (!ok) {
            if (dx) (void)hipFree(dx);
            if (du) (void)hipFree(du);
            if (px) (void)hipHostFree(px);
            if (pu) (void)hipHostFree(pu);
            return NMPC_EHIP;
        }
        h->d_xs = dx; h->d_us = du; h->h_px = px; h->h_pu = pu;
    }
    const size_t n = (size_t)B * N1;
    hipLaunchKernelGGL(nmpc_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_traj, B, (int)N,
                       h->d_xs, h->d_us);
    HIP_TRY(hipGetLastError());
    // The transfers are issued on an idle stream: enqueued behind a solve that is still running, the same two
    // copies measured 1.5x slower end to end (2.9 M against 4.25 M trajectories/s at B = 4096) -- the runtime
    // picks its copy path when the call is made.  The callers wait for the stream anyway.
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_px, h->d_xs, sizeof(double) * n * 6, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_pu, h->d_us, sizeof(double) * (size_t)B * N * 2, hipMemcpyDeviceToHost, h->stream));
    return NMPC_OK;
}

int nmpc_get(nmpc_handle *h, int B, double *x, double *u, int32_t *status, double *cost,
             int32_t *ipm_iters, int32_t *sqp_iters)
{
    if (!h || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    const size_t N = (size_t)h->cfg.N, N1 = N + 1;
    if (x || u) {
        const int rc = stage_trajectories(h, B);
        if (rc != NMPC_OK) return rc;
    }
    if (status) HIP_TRY(hipMemcpyAsync(status, h->d_status, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    if (cost) HIP_TRY(hipMemcpyAsync(cost, h->d_cost, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    if (ipm_iters) HIP_TRY(hipMemcpyAsync(ipm_iters, h->d_iters, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    if (sqp_iters) HIP_TRY(hipMemcpyAsync(sqp_iters, h->d_sqp, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (x) memcpy(x, h->h_px, sizeof(double) * (size_t)B * N1 * 6);
    if (u) memcpy(u, h->h_pu, sizeof(double) * (size_t)B * N * 2);
    return NMPC_OK;
}

int nmpc_get_view(nmpc_handle *h, int B, const double **x, const double **u)
{
    if (!h || !x || !u || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    const int rc = stage_trajectories(h, B);
    if (rc != NMPC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    *x = h->h_px;
    *u = h->h_pu;
    return NMPC_OK;
}

int nmpc_get_controls(nmpc_handle *h, int B, double *ctrl)
{
    if (!h || !ctrl || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(ctrl, h->out_ctrl, sizeof(double) * (size_t)B * 6, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NMPC_OK;
}

// ---- path front-end and closed loop

int nmpc_set_track(nmpc_handle *h, const double *xy, int n_points)
{
    if (!h || !xy || n_points < 5) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t cap = (size_t)h->capacity, N1 = (size_t)h->cfg.N + 1;
    // all or nothing: a failed allocation leaves the handle without a track, never with a partial set
    // of front-end buffers (the kernels of nmpc_frontend / nmpc_closed_loop take all five)
    double *trk = nullptr;
    HIP_TRY(hipMalloc((void **)&trk, sizeof(double) * 2 * (size_t)n_points));
    if (!h->d_ws) {
        struct { double **p; size_t bytes; } fe[] = {
            {&h->d_ws, sizeof(double) * 6 * cap}, {&h->d_pv, sizeof(double) * 2 * cap},
            {&h->d_rot, sizeof(double) * cap}, {&h->d_xy, sizeof(double) * 2 * cap * N1},
            {&h->d_wsfe, sizeof(double) * 6 * cap}};
        bool ok = true;
        for (auto &a : fe)
            if (ok) ok = hip_ok(hipMalloc((void **)a.p, a.bytes), "hipMalloc");
        if (!ok) {
            for (auto &a : fe) { if (*a.p) (void)hipFree(*a.p); *a.p = nullptr; }
            (void)hipFree(trk);
            return NMPC_EHIP;
        }
    }
    if (h->d_track) (void)hipFree(h->d_track);
    h->d_track = trk;
    HIP_TRY(hipMemcpyAsync(h->d_track, xy, sizeof(double) * 2 * (size_t)n_points, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->n_track = n_points;
    return NMPC_OK;
}

static int launch_frontend(nmpc_handle *h, int B)
{
    hipLaunchKernelGGL(nmpc_frontend_kernel, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->d_track, h->n_track,
                       h->d_ws, h->d_pv, B, h->d_x0, h->d_p, h->d_rot, h->d_wsfe);
    HIP_TRY(hipGetLastError());
    h->in_x0 = h->d_x0;
    h->in_p = h->d_p;
    return NMPC_OK;
}

int nmpc_frontend(nmpc_handle *h, const double *world_state, const double *pitch_vref, int B)
{
    if (!h || !world_state || !pitch_vref || B <= 0 || B > h->capacity || !h->d_track) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->d_ws, world_state, sizeof(double) * 6 * (size_t)B, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_pv, pitch_vref, sizeof(double) * 2 * (size_t)B, hipMemcpyHostToDevice, h->stream));
    const int rc = launch_frontend(h, B);
    if (rc != NMPC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));   // the caller's buffers may be reused
    return NMPC_OK;
}

int nmpc_get_frontend(nmpc_handle *h, int B, double *x0_solver, double *params, double *rotation)
{
    if (!h || B <= 0 || B > h->capacity || !h->d_track) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    if (x0_solver) HIP_TRY(hipMemcpyAsync(x0_solver, h->d_x0, sizeof(double) * 6 * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    if (params) HIP_TRY(hipMemcpyAsync(params, h->d_p, sizeof(double) * 6 * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    if (rotation) HIP_TRY(hipMemcpyAsync(rotation, h->d_rot, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NMPC_OK;
}

int nmpc_get_world_path(nmpc_handle *h, int B, double *xy)
{
    if (!h || !xy || B <= 0 || B > h->capacity || !h->d_track) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    const int N1 = h->cfg.N + 1;
    // with the pose the front-end transformed, not the one the plant of nmpc_closed_loop has moved on to
    hipLaunchKernelGGL(nmpc_unrotate_kernel, dim3((B * N1 + 255) / 256), dim3(256), 0, h->stream, h->d_traj, h->d_wsfe,
                       h->d_rot, B, N1, h->d_xy);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(xy, h->d_xy, sizeof(double) * 2 * (size_t)B * N1, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NMPC_OK;
}

static int closed_loop_impl(nmpc_handle *h, int B, int steps, double dt_plant, double steering_ratio,
                            double *world_state, const double *pitch_vref, const double *steer_meas,
                            double *commands, double *trace, int32_t *n_failed)
try {
    if (!h || !world_state || !pitch_vref || B <= 0 || B > h->capacity || steps < 1 || !h->d_track ||
        !(dt_plant > 0.0) || !(steering_ratio != 0.0))
        return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->d_ws, world_state, sizeof(double) * 6 * (size_t)B, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_pv, pitch_vref, sizeof(double) * 2 * (size_t)B, hipMemcpyHostToDevice, h->stream));
    // per-tick rows in and out (measured steering-wheel angles, published commands) go through d_xy, the
    // un-rotation buffer: [cap][N+1][2] doubles, idle during the loop
    double *d_row = h->d_xy, *d_cmd = h->d_xy + (size_t)h->capacity;
    int failed = 0;
    std::vector<int32_t> st;
    for (int s = 0; s < steps; ++s) {
        // (measurement ->) front-end -> one SQP-RTI iteration (warm start, no shift) -> actuation + plant,
        // all on the handle's stream; no host round trip unless the caller asked for a trace
        if (steer_meas) {
            HIP_TRY(hipMemcpyAsync(d_row, steer_meas + (size_t)s * B, sizeof(double) * (size_t)B, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(nmpc_measured_steering_kernel, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->d_ws, d_row, B,
                               steering_ratio);
            HIP_TRY(hipGetLastError());
        }
        int rc = launch_frontend(h, B);
        if (rc != NMPC_OK) return rc;
        rc = nmpc_solve(h, B, NMPC_MODE_RTI, 1, 0.0);
        if (rc != NMPC_OK) return rc;
        hipLaunchKernelGGL(nmpc_plant_kernel, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->d_ws, h->d_pv, h->out_ctrl,
                           B, dt_plant, 1.0 / h->cfg.wheelbase, steering_ratio, steer_meas ? 1 : 0,
                           commands ? d_cmd : (double *)nullptr);
        HIP_TRY(hipGetLastError());
        if (commands)
            HIP_TRY(hipMemcpyAsync(commands + (size_t)s * B * 2, d_cmd, sizeof(double) * 2 * (size_t)B,
                                   hipMemcpyDeviceToHost, h->stream));
        if (trace)
            HIP_TRY(hipMemcpyAsync(trace + (size_t)s * B * 6, h->d_ws, sizeof(double) * 6 * (size_t)B,
                                   hipMemcpyDeviceToHost, h->stream));
        if (n_failed) {
            st.resize((size_t)B);
            HIP_TRY(hipMemcpyAsync(st.data(), h->d_status, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            for (int32_t v : st) failed += v != 0;
        }
    }
    HIP_TRY(hipMemcpyAsync(world_state, h->d_ws, sizeof(double) * 6 * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (n_failed) *n_failed = failed;
    return NMPC_OK;
} catch (const std::bad_alloc &) {   // host containers: no exception may cross the C ABI
    snprintf(g_err, sizeof(g_err), "nmpc_closed_loop: out of host memory");
    return NMPC_ENOMEM;
} catch (...) {
    snprintf(g_err, sizeof(g_err), "nmpc_closed_loop: unexpected exception");
    return NMPC_EINVAL;
}

int nmpc_closed_loop(nmpc_handle *h, int B, int steps, double dt_plant, double steering_ratio,
                     double *world_state, const double *pitch_vref, double *trace, int32_t *n_failed)
{
    return closed_loop_impl(h, B, steps, dt_plant, steering_ratio, world_state, pitch_vref, nullptr, nullptr, trace, n_failed);
}

int nmpc_closed_loop_measured(nmpc_handle *h, int B, int steps, double dt_plant, double steering_ratio,
                              double *world_state, const double *pitch_vref, const double *steer_meas,
                              double *commands, double *trace, int32_t *n_failed)
{
    if (!steer_meas) return NMPC_EINVAL;
    return closed_loop_impl(h, B, steps, dt_plant, steering_ratio, world_state, pitch_vref, steer_meas, commands, trace, n_failed);
}

// ---- multi-GPU: all-gather of the control records over RCCL.  RCCL is bound at
// run time (dlopen), so the library has no link-time dependency on it and single-GPU users never load it.
typedef int (*nccl_all_gather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
static nccl_all_gather_fn g_all_gather = nullptr;
static std::mutex g_rccl_mu;

static int bind_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_all_gather) return NMPC_OK;
    void *lib = nullptr;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) {
        snprintf(g_err, sizeof(g_err), "nmpc_gather_controls: cannot load librccl.so (%s)", dlerror());
        return NMPC_EHIP;
    }
    g_all_gather = (nccl_all_gather_fn)dlsym(lib, "ncclAllGather");
    if (!g_all_gather) {
        snprintf(g_err, sizeof(g_err), "nmpc_gather_controls: ncclAllGather not found in librccl.so");
        return NMPC_EHIP;
    }
    return NMPC_OK;
}

int nmpc_gather_controls(nmpc_handle *h, void *nccl_comm, int B, double *d_all)
{
    if (!h || !nccl_comm || !d_all || B <= 0 || B > h->capacity) return NMPC_EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    const int rc = bind_rccl();
    if (rc != NMPC_OK) return rc;
    const int ncclFloat64 = 8;      // rccl.h: ncclFloat64 = ncclDouble = 8
    const int st = g_all_gather(h->out_ctrl, d_all, (size_t)B * 6, ncclFloat64, nccl_comm, h->stream);
    if (st != 0) {
        snprintf(g_err, sizeof(g_err), "nmpc_gather_controls: ncclAllGather returned %d", st);
        return NMPC_EHIP;
    }
    return NMPC_OK;
}

void *nmpc_device_ptr(nmpc_handle *h, int which)
{
    if (!h) return nullptr;
    switch (which) {
        case 0: return h->out_ctrl;
        case 1: return h->d_traj;
        case 2: return h->d_status;
        default: return nullptr;
    }
}

void *nmpc_stream(nmpc_handle *h) { return h ? (void *)h->stream : nullptr; }
int nmpc_capacity(const nmpc_handle *h) { return h ? h->capacity : NMPC_EINVAL; }
int nmpc_horizon(const nmpc_handle *h) { return h ? h->cfg.N : NMPC_EINVAL; }

}  // extern "C"
