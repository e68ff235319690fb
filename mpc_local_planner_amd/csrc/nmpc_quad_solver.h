// nmpc_quad_solver.h -- SQP-RTI solves of FOUR path-tracking OCP instances by one workgroup of
// four 64-lane wavefronts on gfx950 (horizons N <= 63).  Body of nmpc_quad_kernel in nmpc_hip.hip;
// like nmpc_wave_solver.h (whose model code, LDS stage layout and vector-sweep stages it shares) it
// is written against a wave policy W so that tests run the same source on the CPU thread emulator.
//
// What it replaces (reference file:line): the same path as nmpc_wave_solver.h --
//   Acados::Solver::solve src/AcadosSolver.cpp:28-76 -> path_tracking_acados_solve (SQP-RTI)
//   src/AcadosPathTracking.cpp:26 [acados/HPIPM, not in tree]; model / cost / bounds
//   acados/mpcPathTracking.py:41-47, 63-80, 100-115.
//
// Why four instances: the serial Riccati sweeps use few lanes (a 9-column matrix row per
// instruction), and the one cross-lane operation fp64 has on gfx950 -- v_fmac_f64 with a DPP
// row broadcast -- is confined to a 16-lane row.  So one wave sweeps four instances at once, one
// per 16-lane row, with the recursion matrices held in REGISTERS across stages:
//   * element-wise work (stage layout): wave w owns instance w of the workgroup, lane k = shooting
//     node k: ERK4 + sensitivities, Gauss-Newton cost, slacks / multipliers / step lengths.
//   * sweeps: ONE wave of the workgroup (the sweeper) runs the factor sweep, the forward sweeps and
//     the corrector's backward sweep for all four instances; row r of its lanes = instance r,
//     lane j of a row = column j of the augmented stage matrices [u0 u1 X Y psi v delta thr | 1].
//     Factor stage, all on v_fmac_f64_dpp (no LDS round trip on the serial path):
//         Mt_i  = Pt_i e8 + sum_m bcast(Pt_i, m+2) BA_m            (42 instr, i = 0..5 rows of Pt)
//         Ht_i  = C_i + sum_m bcast(BA_m, i) Mt_m                  (27 instr, structure of [B A] used)
//         Schur complement on the 2x2 input block -> K, kff, -inv(Huu), new Pt   (27 instr)
//     where register X_i holds row i of a matrix and lane j its column j; lane 8 is the augmented
//     column (gap b / gradient g / cost-to-go gradient p), lanes 9,10 carry unit columns through the
//     elimination and so deliver the rows of -inv(Huu).
//   The four waves meet at four workgroup barriers per interior-point iteration.  Each wave is a SLOT:
//   when its instance has terminated it writes the result back and draws the next instance from a
//   launch-wide ticket queue, so no instance waits for the slowest of a group of four.
//
// Status (measured on MI355X, profiles/r2): per solve this kernel issues 4.0x fewer
// LDS instructions (19.7 k -> 5.0 k), keeps the LDS array busy 3.8x less (68 k -> 18 k cycles) and issues
// 44 % fewer VALU instructions than the one-instance-per-workgroup kernel.  That buys 3-7 % of throughput,
// not the factor the counters suggest: LDS capacity (19 KB per instance at N = 40) and 256 VGPRs per slot
// wave both stop at eight instances per CU, so a CU runs two sweep streams where the other kernel runs
// eight waves, and an interior-point iteration of a workgroup (~24-30 us) costs what it costs there.
// nmpc_solve picks it where it wins: 32 <= N <= 43 (two workgroups per CU) and batches of 8192 and more
// (nmpc_hip.hip: use_quad_kernel; NMPC_KERNEL=quad / wave forces a kernel).
#pragma once

#include "nmpc_wave_solver.h"

namespace nmpcq {

using namespace nmpcw;

constexpr int QI = 4;            // instances per workgroup = 16-lane rows per wave
#ifndef NMPC_QUAD_SYM
#define NMPC_QUAD_SYM 8          // the cost-to-go is symmetrised every so many stages of the factor sweep (even)
#endif
// per-stage constants of the factor sweep in the spare slot (offset 5) of the column groups, so that
// every lane of every start-value load walks the stages with the same stride
constexpr int S_Q11 = 0 * GS + 5, S_Q22 = 1 * GS + 5, S_Q33 = 2 * GS + 5, S_ZERO = 3 * GS + 5, S_ONE = 4 * GS + 5;
// LDS region of one instance, padded so that regions of neighbouring instances start 16 doubles apart
// modulo the 32 double-wide bank set (the two instances that share a 32-lane LDS access group then
// read different banks when their lanes read the same offsets)
constexpr int inst_stride(int N)
{
    const int n = (N + 1) * ST + X_SIZE;
    return n + ((16 - n % 32) + 32) % 32;
}
constexpr int WG_FLAGS = 8;      // workgroup area after the four regions: done flags
#ifdef NMPC_QUAD_TIMELINE
constexpr int WG_TIMELINE = 64;   // debug build: per-wave phase accumulators behind the flags
#else
constexpr int WG_TIMELINE = 0;
#endif
constexpr int quad_lds_doubles(int N) { return QI * inst_stride(N) + WG_FLAGS + WG_TIMELINE; }

// start values of the eight accumulators Ht_i of a factor stage: offset (doubles, in the stage) of the
// entry lane j loads for row i.  Cost entries on and next to the diagonal, the gradient in the
// augmented column (lane 8), unit columns for the two input rows (lanes 9, 10), zero elsewhere.
NMPC_DEV int hinit_offset(int i, int j)
{
    if (j == 8) return I_G + i;
    switch (i) {
        case 0: return j == 0 ? I_GAM01 : j == 9 ? S_ONE : S_ZERO;
        case 1: return j == 1 ? I_GAM01 + 1 : j == 10 ? S_ONE : S_ZERO;
        case 2: return j == 2 ? I_Q0 : j == 3 ? I_Q0 + 1 : j == 4 ? I_Q2 : S_ZERO;
        case 3: return j == 2 ? I_Q0 + 1 : j == 3 ? S_Q11 : S_ZERO;
        case 4: return j == 2 ? I_Q2 : j == 4 ? S_Q22 : S_ZERO;
        case 5: return j == 5 ? S_Q33 : S_ZERO;
        case 6: return j == 6 ? I_GAM23 : S_ZERO;
        default: return j == 7 ? I_GAM23 + 1 : S_ZERO;
    }
}

#define NMPC_ZQ(a, q) ((q) == 0 ? (a)[0] : (q) == 1 ? (a)[1] : (q) == 2 ? (a)[6] : (a)[7])
#define NMPC_VCOL(j) ((j) < 2 ? (j) : (j) - 2)
#define NMPC_KOFF(c) ((c) < 2 ? O_E + 2 * (c) : (c) < 6 ? GS * (c) + G_K : O_KFF)

// The batch one launch works on (device pointers; the layout of include/nmpc.h).
struct Batch {
    int B;                      // instances 0..B-1
    const double *x0, *p, *W, *bnd;
    int w_stride, bnd_stride;   // doubles between the rows of W / bnd (0: one shared row)
    double *traj, *ctrl, *cost;
    int *status, *iters, *sqp_iters;
    unsigned char *reset;       // cold-start flags, in/out
    const int *order;           // ticket -> instance (longest expected solve first), or null
    int *queue;                 // next ticket; zero at launch
};

// Every wave of the workgroup is a SLOT that holds one instance at a time: when its instance has
// terminated it writes the result back and draws the next ticket from the launch-wide queue (one
// atomic add), so the four rows of the sweeps stay busy until the batch is exhausted -- no instance
// waits for the slowest of its group, and the launch has no rounds of workgroups with a ragged end.
// The workgroup ends when the queue is empty and its four slots are.
template <class W>
NMPC_DEV void solve_quad(W &w, const Problem &pb, const Batch &bt)
{
    const int lane = w.lane();          // element-wise phases: shooting node; sweeps: (row, column)
    const int wv = w.wave();
    const int N = pb.N;
    const double h = pb.dt;
    const int IS = inst_stride(N);
    double *const ct = w.lds() + wv * IS;       // pseudo-stage -1 of this wave's slot
    double *const stg = ct + ST;
    double *const xa = ct + (N + 1) * ST;
    double *const flags = w.lds() + QI * IS;

    const bool has_node = lane <= N;
    const bool act = lane < N;
    const bool actx = act && lane >= 1;
    const int node = has_node ? lane : N;
    const bool mq[4] = {act, act, actx, actx};
    const double inv_mcount = w.uniform(w.rcp((double)(4 * N + 4 * (N - 1))));

    // state of the slot's instance (stage layout: this lane's node)
    InstanceIO io = {};
    int inst = 0;
    bool have = false, idle = false;    // uniform over the wave
    double xb[6] = {0, 0, 0, 0, 0, 0}, ub_[2] = {0, 0};
    double lb[4] = {0, 0, 0, 0}, ub[4] = {0, 0, 0, 0};
    double R00 = 0, R11 = 0, Q11 = 0, Q22 = 0, Q33 = 0, Q44 = 0, Q55 = 0;
    double rb[6] = {0, 0, 0, 0, 0, 0}, g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Q00 = 0, Q01 = 0, Q02 = 0;
    double cost_out = 0.0;
    double cur[4] = {0, 0, 0, 0};
    double tl[4] = {1, 1, 1, 1}, tu[4] = {1, 1, 1, 1}, ll[4] = {0, 0, 0, 0}, lu[4] = {0, 0, 0, 0};
    double z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double lin_scale = 1.0;
    int it = 0;

    // constant entries of [B A] (row v): functions of dt only, so any state gives them -- the sweeper
    // needs them even if its own slot never receives an instance
    double a33, a35, b31;
    {
        double phi[6], aX[6], aY[6], aP[6], aV[6];
        rk4_sens(pb, xb, ub_, 0.0, phi, aX, aY, aP, aV);
        a33 = w.uniform(aV[1]); a35 = w.uniform(aV[3]); b31 = w.uniform(aV[5]);
    }

    // per-lane view of the sweeps (only the sweeper wave uses it): row = instance, j = column.  The
    // addresses derived from it are set up per pass (opaque lane index), not held through the slot loop
    const int j16 = lane & 15;

    // (w.stamp: phase timeline of the debug build NMPC_QUAD_TIMELINE, a no-op otherwise; segment = time since
    // the previous stamp: 0 slot work + E-A, 1 barrier 1, 2 factor sweep, 3 forward sweep, 4 barrier 2,
    // 5 E-B, 6 barrier 3, 7 backward sweep, 8 forward sweep, 9 barrier 4, 10 E-C, 11 one tick per iteration)
    w.stamp_init(w.lds() + QI * IS + WG_FLAGS);
    for (;;) {
        double mu = 0.0;
        // ---- the slot: finish an instance that has terminated, fetch the next one
        for (;;) {
            if (!have) {
                if (idle) break;
                const int ticket = w.next_ticket(bt.queue);
                if (ticket >= bt.B) { idle = true; break; }
                inst = bt.order ? bt.order[ticket] : ticket;
                io.x0 = bt.x0 + (size_t)inst * 6;
                io.p = bt.p + (size_t)inst * 6;
                io.W = bt.W + (size_t)inst * bt.w_stride;
                io.bnd = bt.bnd + (size_t)inst * bt.bnd_stride;
                io.traj = bt.traj + (size_t)inst * (N + 1) * 8;
                io.ctrl = bt.ctrl + (size_t)inst * 6;
                io.status = bt.status + inst;
                io.iters = bt.iters + inst;
                io.sqp_iters = bt.sqp_iters + inst;
                io.cost = bt.cost + inst;
                io.cold = bt.reset[inst];
                // ---- iterate (stage layout)
                {
                    const double *t = io.traj + node * 8;
#pragma unroll
                    for (int m = 0; m < 6; ++m) xb[m] = io.cold ? io.x0[m] : t[m];
                    ub_[0] = (io.cold || !act) ? 0.0 : t[6];
                    ub_[1] = (io.cold || !act) ? 0.0 : t[7];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { lb[q] = io.bnd[q]; ub[q] = io.bnd[4 + q]; }
                R00 = w.uniform(h * io.W[5]); R11 = w.uniform(h * io.W[6]); Q11 = w.uniform(h * io.W[0]);
                Q22 = w.uniform(h * io.W[1]); Q33 = w.uniform(h * io.W[2]); Q44 = w.uniform(h * io.W[3]);
                Q55 = w.uniform(h * io.W[4]);
                // ---- linearise (stage layout)
                {
                    double gp, gpc;
                    sincos_bounded(io.p[4], &gp, &gpc);
                    gp *= 9.81;
                    double phi[6], aX[6], aY[6], aP[6], aV[6];
                    // (step size opaque per instance: what depends on it alone -- the v-row of the
                    // sensitivities -- would otherwise be computed ahead of the slot loop and spilled)
                    Problem pl = pb;
                    pl.dt = w.rounded(pb.dt);
                    rk4_sens(pl, xb, ub_, gp, phi, aX, aY, aP, aV);
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        const double xn = w.shfl_down1(xb[m]);
                        rb[m] = w.rounded(act ? (w.rounded(phi[m]) - xn) : 0.0);
                    }
                    if (lane == 0) {   // fold the known dx_0 = x0 - x_0 into the first gap: b_0 += A_0 dx0
                        double d[6];
#pragma unroll
                        for (int m = 0; m < 6; ++m) d[m] = io.x0[m] - xb[m];
                        rb[0] += d[0] + aX[0] * d[2] + aX[1] * d[3] + aX[2] * d[4] + aX[3] * d[5];
                        rb[1] += d[1] + aY[0] * d[2] + aY[1] * d[3] + aY[2] * d[4] + aY[3] * d[5];
                        rb[2] += aP[0] * d[2] + aP[1] * d[3] + aP[2] * d[4] + aP[3] * d[5];
                        rb[3] += aV[1] * d[3] + aV[3] * d[5];
                        rb[4] += d[4];
                        rb[5] += d[5];
                    }
                    if (act) {   // rows X,Y,psi of [B A] into the column groups; per-stage constants beside them
                        double *c = stg + lane * ST + G_XY;
                        st2(c + 0 * GS, aX[4], aY[4]); st2(c + 0 * GS + 2, aP[4], Q11);
                        st2(c + 1 * GS, aX[5], aY[5]); st2(c + 1 * GS + 2, aP[5], Q22);
                        st2(c + 2 * GS, aX[0], aY[0]); st2(c + 2 * GS + 2, aP[0], Q33);
                        st2(c + 3 * GS, aX[1], aY[1]); st2(c + 3 * GS + 2, aP[1], 0.0);
                        st2(c + 4 * GS, aX[2], aY[2]); st2(c + 4 * GS + 2, aP[2], 1.0);
                        st2(c + 5 * GS, aX[3], aY[3]); st2(c + 5 * GS + 2, aP[3], 0.0);
                        st2(c - G_XY + 0 * GS + G_K, 0.0, 0.0);
                        st2(c - G_XY + 1 * GS + G_K, 0.0, 0.0);
                    }
                    if (lane == 0) {
                        double *c = ct + G_XY;          // rows v,delta,thr in the (X,Y,psi) positions of each group
                        c[0 * GS] = 0.0; c[0 * GS + 1] = h;   c[0 * GS + 2] = 0.0;     // u0
                        c[1 * GS] = b31; c[1 * GS + 1] = 0.0; c[1 * GS + 2] = h;       // u1
                        c[2 * GS] = 0.0; c[2 * GS + 1] = 0.0; c[2 * GS + 2] = 0.0;     // psi
                        c[3 * GS] = a33; c[3 * GS + 1] = 0.0; c[3 * GS + 2] = 0.0;     // v
                        c[4 * GS] = 0.0; c[4 * GS + 1] = 1.0; c[4 * GS + 2] = 0.0;     // delta
                        c[5 * GS] = a35; c[5 * GS + 1] = 0.0; c[5 * GS + 2] = 1.0;     // thr
                        double *u = xa + X_UNIT;
                        u[0] = 1.0; u[1] = 0.0; u[2] = 0.0; u[3] = 0.0; u[4] = 0.0; u[5] = 1.0; u[6] = 0.0; u[7] = 0.0;
                        double *ub2 = xa + X_UNITB;
                        ub2[0] = 1.0; ub2[1] = 0.0; ub2[2] = 0.0; ub2[3] = 0.0; ub2[4] = 0.0; ub2[5] = 1.0; ub2[6] = 0.0;
                        double *zz = xa + X_ZERO;
                        zz[0] = 0.0; zz[1] = 0.0; zz[2] = 0.0; zz[3] = 0.0;
                        double *ft = xa + X_FT;
                        ft[0] = 1.0; ft[1] = 0.0; ft[2] = 0.0; ft[3] = 0.0; ft[4] = 1.0; ft[5] = 0.0;
                    }
                    // Gauss-Newton cost terms (mpcPathTracking.py:63-80,103)
                    const double c0 = io.p[0], c1 = io.p[1], c2 = io.p[2], c3 = io.p[3];
                    const double X = xb[0];
                    const double s = 3.0 * c3 * X * X + 2.0 * c2 * X + c1;
                    const double dsl = 6.0 * c3 * X + 2.0 * c2;
                    const double ypath = c3 * X * X * X + c2 * X * X + c1 * X + c0;
                    const double cte = ypath - xb[1];
                    const double epsi = xb[2] - atan(s);
                    const double d = -dsl / (1.0 + s * s);
                    const double ev = xb[3] - io.p[5];
                    Q00 = Q11 * s * s + Q22 * d * d;
                    Q01 = -Q11 * s;
                    Q02 = Q22 * d;
                    g[0] = R00 * ub_[0];
                    g[1] = R11 * ub_[1];
                    g[2] = Q11 * s * cte + Q22 * d * epsi;
                    g[3] = -Q11 * cte;
                    g[4] = Q22 * epsi;
                    g[5] = Q33 * ev;
                    g[6] = Q44 * xb[4];
                    g[7] = Q55 * xb[5];
                    const double cst = 0.5 * (Q11 * cte * cte + Q22 * epsi * epsi + Q33 * ev * ev +
                                              Q44 * xb[4] * xb[4] + Q55 * xb[5] * xb[5] +
                                              R00 * ub_[0] * ub_[0] + R11 * ub_[1] * ub_[1]);
                    cost_out = w.sum(act ? cst : 0.0);
                }
                // ---- interior point: starting point
                cur[0] = ub_[0]; cur[1] = ub_[1]; cur[2] = xb[4]; cur[3] = xb[5];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double a = fmax(cur[q] - lb[q], pb.thr0), b = fmax(ub[q] - cur[q], pb.thr0);
                    tl[q] = mq[q] ? a : 1.0;
                    tu[q] = mq[q] ? b : 1.0;
                    ll[q] = mq[q] ? pb.mu0 * w.rcp(a) : 0.0;
                    lu[q] = mq[q] ? pb.mu0 * w.rcp(b) : 0.0;
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) z[m] = 0.0;
                lin_scale = 1.0;
                it = 0;
                have = true;
            }
            // ---- E-A: termination test of this instance
            int qp_status = -1;
            {
                double mu_loc = 0.0, c_loc = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    mu_loc += tl[q] * ll[q] + tu[q] * lu[q];
                    c_loc = fmax(c_loc, fmax(fmin(tl[q], ll[q]), fmin(tu[q], lu[q])));
                }
                mu = w.sum(mu_loc) * inv_mcount;
                const bool near_end = it > 0 && mu <= pb.tol_mu && lin_scale <= pb.tol_lin;
                const double cmax = near_end ? w.max(c_loc) : 1.0;
                if (!(mu == mu)) qp_status = 1;
                else if (near_end && cmax <= pb.tol_comp) qp_status = 0;
                else if (it >= pb.ipm_max_iter) qp_status = (mu <= 1e-8 && lin_scale <= 1e-8) ? 0 : 4;
                else if (pb.stall_iter > 0 && it >= pb.stall_iter && lin_scale > pb.stall_lin) qp_status = 4;   // stalled
            }
            if (qp_status < 0) break;       // not finished: this round iterates on it
            // ---- full step (alpha = 1, no line search), write back, free the slot
            {
                bool bad = false;
#pragma unroll
                for (int m = 0; m < 8; ++m) bad = bad || !(fabs(z[m]) <= 1e300);
                const double nbad = w.sum((bad && has_node) ? 1.0 : 0.0);
                if (qp_status == 0 && nbad != 0.0) qp_status = 1;
                // the iterate of this node again (only its bounded variables stayed in registers)
                {
                    const double *t = io.traj + node * 8;
#pragma unroll
                    for (int m = 0; m < 4; ++m) xb[m] = io.cold ? io.x0[m] : t[m];
                    xb[4] = cur[2]; xb[5] = cur[3];
                    ub_[0] = cur[0]; ub_[1] = cur[1];
                }
                if (qp_status != 1) {
                    ub_[0] += z[0];
                    ub_[1] += z[1];
#pragma unroll
                    for (int m = 0; m < 6; ++m) xb[m] += z[2 + m];
                    if (lane == 0) {
#pragma unroll
                        for (int m = 0; m < 6; ++m) xb[m] = io.x0[m];
                    }
                }
                const double u00 = w.readlane(ub_[0], 0), u01 = w.readlane(ub_[1], 0);
                const double d1 = w.readlane(xb[4], 1), t1 = w.readlane(xb[5], 1);
                if (has_node) {
                    double *t = io.traj + lane * 8;
#pragma unroll
                    for (int m = 0; m < 6; ++m) t[m] = xb[m];
                    t[6] = act ? ub_[0] : 0.0;
                    t[7] = act ? ub_[1] : 0.0;
                }
                if (lane == 0) {
                    io.ctrl[0] = u00; io.ctrl[1] = u01; io.ctrl[2] = d1; io.ctrl[3] = t1;
                    io.ctrl[4] = cost_out;
                    io.ctrl[5] = (double)(qp_status + 16 * it);
                    *io.status = qp_status;
                    *io.iters = it;
                    *io.sqp_iters = 1;
                    *io.cost = cost_out;
                    // instances that did not return success restart cold next time (the reference
                    // re-initialises on every status != ACADOS_SUCCESS, src/AcadosSolver.cpp:56-59)
                    bt.reset[inst] = qp_status != 0 ? 1 : 0;
                }
                have = false;
            }
        }
        const bool done = !have;        // nothing to iterate on in this slot (uniform over the wave)
        if (!done) {
            double G[8];
            G[0] = R00 * z[0] + g[0];
            G[1] = R11 * z[1] + g[1];
            G[2] = Q00 * z[2] + Q01 * z[3] + Q02 * z[4] + g[2];
            G[3] = Q01 * z[2] + Q11 * z[3] + g[3];
            G[4] = Q02 * z[2] + Q22 * z[4] + g[4];
            G[5] = Q33 * z[5] + g[5];
            G[6] = Q44 * z[6] + g[6];
            G[7] = Q55 * z[7] + g[7];
            double gam[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double wv_ = NMPC_ZQ(z, q);
                const double rtl = mq[q] ? (wv_ - (lb[q] - cur[q]) - tl[q]) : 0.0;
                const double rtu = mq[q] ? ((ub[q] - cur[q]) - wv_ - tu[q]) : 0.0;
                const double rl = ll[q] * w.rcp(tl[q]), ru = lu[q] * w.rcp(tu[q]);
                gam[q] = rl + ru;
                const double gq = rl * rtl - ru * rtu;
                if (q == 0) G[0] += gq;
                if (q == 1) G[1] += gq;
                if (q == 2) G[6] += gq;
                if (q == 3) G[7] += gq;
            }
            if (act) {
                double *sl = stg + lane * ST;
                // the complete diagonal entries of the bounded variables (cost + barrier)
                st2(sl + I_GAM01, R00 + gam[0], R11 + gam[1]); st2(sl + I_GAM23, Q44 + gam[2], Q55 + gam[3]);
                st2(sl + I_Q0, Q00, Q01);                      st2(sl + I_Q2, Q02, 0.0);
#pragma unroll
                for (int m = 0; m < 8; m += 2) st2(sl + I_G + m, G[m], G[m + 1]);
                st2(sl + O_D + 0, rb[0], rb[1]); st2(sl + O_D + 2, rb[2], 0.0);
                st2(sl + O_D + 4, rb[3], rb[4]); st2(sl + O_D + 6, rb[5], 0.0);
            }
        }
        if (lane == 0) flags[wv] = done ? 1.0 : 0.0;
        w.stamp(0);
        w.sync();                                                         // barrier 1
        w.stamp(1);
        // the queue is empty and no slot holds work: the workgroup ends (uniform over all four waves)
        const bool all_done = w.uniform_flag(flags[0] != 0.0 && flags[1] != 0.0 && flags[2] != 0.0 && flags[3] != 0.0);
        if (all_done) { w.stamp_flush(); break; }
        w.tick(11);

        double dzs[8];
        double smu = 0.0, csc = 0.0, mu_aff = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (w.sweeper()) {
                w.sweep_priority(true);
                const int row = w.opaque(lane) >> 4;
                double *const rct = w.lds() + row * IS;
                double *const rstg = rct + ST;
                double *const rxa = rct + (N + 1) * ST;
                double *const rlast = rstg + (N - 1) * ST;
                if (pass == 0) {
                    // ---- factor sweep of the four instances, k = N-1 .. 0
                    const int j = w.opaque(j16);
                    const double *tj, *bj;       // column j of [B A b]: rows X,Y,psi / rows v,delta,thr
                    int sTop, sBot;
                    if (j == 2 || j == 3) { tj = rxa + X_UNIT + 4 * (j - 2); bj = rxa + X_ZERO; sTop = 0; sBot = 0; }
                    else if (j < 8) { tj = rlast + GS * NMPC_VCOL(j) + G_XY; bj = rct + GS * NMPC_VCOL(j) + G_XY; sTop = ST; sBot = 0; }
                    else if (j == 8) { tj = rlast + O_D; bj = rlast + O_D + 4; sTop = ST; sBot = ST; }
                    else { tj = rxa + X_ZERO; bj = rxa + X_ZERO; sTop = 0; sBot = 0; }
                    // start values of Ht: one address per accumulator, pointing at the LOWER stage of the
                    // pair of stages one trip of the loop handles (the upper one is an immediate offset away)
                    const double *ha[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) ha[i] = rlast - ST + hinit_offset(i, j);
                    double *ka = rxa + X_DUM;
                    int sK = 0;
                    if (j >= 2 && j < 8) { ka = rlast + NMPC_KOFF(j - 2); sK = ST; }
                    else if (j == 8) { ka = rlast + O_KFF; sK = ST; }
                    else if (j == 9) { ka = rlast + O_INV0; sK = ST; }
                    else if (j == 10) { ka = rlast + O_INV1; sK = ST; }
                    const double e8 = j == 8 ? 1.0 : 0.0;
                    // cost-to-go rows live in registers: set A holds Pt in [2..7] while set B accumulates
                    // Ht, whose rows 2..7 are the next Pt -- the two sets take turns
                    double RA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, RB[8];
                    double ba[6];
                    w.fac_fetch(tj, bj, ba);
#pragma unroll 1
                    for (int k = N - 1; k >= 0; k -= 2) {
                        tj -= sTop; bj -= sBot;
                        w.template fac_stage<ST>(RA, RB, ba, ha, tj, bj, ka, e8);
                        ka -= sK;
                        if (k == 0) break;
                        tj -= sTop; bj -= sBot;
                        w.template fac_stage<0>(RB, RA, ba, ha, tj, bj, ka, e8);
                        ka -= sK;
#pragma unroll
                        for (int i = 0; i < 8; ++i) ha[i] -= 2 * ST;
                        // Rows of the cost-to-go are formed independently of each other, so round-off makes
                        // it slightly unsymmetric, and the unsymmetric part is not damped by the recursion:
                        // it is carried from stage to stage by the open-loop dynamics, P_a <- A' P_a A, and
                        // on ill-conditioned stage costs (path curvature of the order of 100) grew to 1e-4
                        // of the gains over 40 stages (found by tools/stress_parity.py; the one-instance
                        // kernel forms only the upper triangle and mirrors it).  Every NMPC_QUAD_SYM stages
                        // the 6x6 block is therefore replaced by its symmetric part, through the (otherwise
                        // unused) cost-to-go slots of the exchange area.
                        if ((((N - 1 - k) >> 1) % (NMPC_QUAD_SYM / 2)) == NMPC_QUAD_SYM / 2 - 1) {
                            double *sp = rxa + X_P;
                            const bool in = j >= 2 && j < 8;
                            const int jj = in ? j - 2 : 0;
                            if (in) {
#pragma unroll
                                for (int i = 0; i < 6; ++i) sp[i * 8 + jj] = RA[i + 2];
                            }
                            w.wsync();
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                const double t = sp[jj * 8 + i];
                                RA[i + 2] = in ? 0.5 * (RA[i + 2] + t) : RA[i + 2];
                            }
                            w.wsync();
                        }
                    }
                    w.sweep_drain();
                    w.wsync();
                    w.stamp(2);
                } else {
                    // ---- backward vector sweep of the corrector, four instances
                    const int vl = w.opaque(j16) & 7;
                    const double *tv;
                    int sTopV;
                    if (vl == 2 || vl == 3) { tv = rxa + X_UNITB + 4 * (vl - 2); sTopV = 0; }
                    else { tv = rlast + GS * NMPC_VCOL(vl) + G_XY; sTopV = ST; }
                    const double *dg = rlast + O_D + (vl < 2 ? vl : vl >= 6 ? vl - 4 : 4);
                    const int sDgV = ST;
                    const double *ab = vl >= 2 ? rlast + NMPC_KOFF(vl - 2) : rlast + (vl == 0 ? O_INV0 : O_INV1);
                    const double keep = vl >= 2 ? 1.0 : 0.0;
                    double *kfw = j16 < 2 ? rlast + O_KFF + j16 : rxa + X_DUM;
                    const int sKfw = j16 < 2 ? ST : 0;
                    const double cb0 = vl == 1 ? b31 : vl == 5 ? a33 : vl == 7 ? a35 : 0.0;
                    const double cb1 = vl == 0 ? h : vl == 6 ? 1.0 : 0.0;
                    const double cb2 = vl == 1 ? h : vl == 7 ? 1.0 : 0.0;
                    double res = 0.0;
                    double da[6], db[6];
                    w.bwd_fetch(tv, dg, ab, da);
#pragma unroll 1
                    for (int k = N - 1; k >= 0; k -= 2) {
                        tv -= sTopV; dg -= sDgV; ab -= ST;
                        w.bwd_stage(da, db, tv, dg, ab, kfw, cb0, cb1, cb2, keep, res);
                        kfw -= sKfw;
                        if (k == 0) break;
                        tv -= sTopV; dg -= sDgV; ab -= ST;
                        w.bwd_stage(db, da, tv, dg, ab, kfw, cb0, cb1, cb2, keep, res);
                        kfw -= sKfw;
                    }
                    w.sweep_drain();
                    w.wsync();
                    w.stamp(7);
                }
                {
                    // ---- forward sweep, four instances (lanes 0..5 of a row: rows of dx_{k+1}, lanes 6,7: du_k)
                    const int fl = w.opaque(j16);
                    const int vl = fl & 7;
                    const double *p1;
                    int s1;
                    if (vl < 3) { p1 = rstg + G_XY + vl; s1 = ST; }
                    else if (vl < 6) { p1 = rct + G_XY + (vl - 3); s1 = 0; }
                    else { p1 = rstg + G_K + (vl - 6); s1 = ST; }
                    const double *p3;
                    int s3;
                    if (vl < 6) { p3 = rxa + X_FT + (vl == 0 ? 0 : vl == 1 ? 2 : 1); s3 = 0; }
                    else { p3 = rstg + O_E + (vl - 6); s3 = ST; }
                    const double *p2;
                    int s2;
                    if (vl >= 6) { p2 = rstg + O_KFF + (vl - 6); s2 = ST; }
                    else if (pass == 0) { p2 = rstg + O_D + (vl < 3 ? vl : vl + 1); s2 = ST; }
                    else { p2 = rstg + O_D + 4; s2 = ST; }
                    double *sw = rxa + X_DUM;
                    int sSw = 0;
                    if (fl < 6) { sw = rstg + O_D + 2 + fl; sSw = ST; }
                    else if (fl < 8) { sw = rct + O_D + (fl - 6); sSw = ST; }
                    double val = 0.0;
                    double fa[9], fb[9];
                    w.fwd_fetch(p1, p3, p2, fa);
#pragma unroll 1
                    for (int k = 0; k < N; k += 2) {
                        p1 += s1; p2 += s2; p3 += s3;
                        w.fwd_stage(fa, fb, p1, p3, p2, sw, val);
                        sw += sSw;
                        if (k + 1 == N) break;
                        p1 += s1; p2 += s2; p3 += s3;
                        w.fwd_stage(fb, fa, p1, p3, p2, sw, val);
                        sw += sSw;
                    }
                    w.sweep_drain();
                    if (j16 == 0) {                 // z_0 = (du_0, 0): dx_0 is folded into r_b
#pragma unroll
                        for (int m = 0; m < 6; ++m) rct[O_D + 2 + m] = 0.0;
                    }
                }
                w.sweep_priority(false);
                w.stamp(pass == 0 ? 3 : 8);
            }
            w.sync();                                                     // barriers 2 and 4
            w.stamp(pass == 0 ? 4 : 9);
            if (!done) {
                // ---- stage layout: pick up the direction of this node (slot node-1 holds z_node)
                double dzl[8];
                {
                    const double *src = ct + node * ST + O_D;
#pragma unroll
                    for (int m = 0; m < 8; m += 2) {
                        const D2 v = ld2(src + m);
                        dzl[m] = has_node ? v.x : 0.0;
                        dzl[m + 1] = has_node ? v.y : 0.0;
                    }
                    if (!act) { dzl[0] = 0.0; dzl[1] = 0.0; }
                }
                if (pass == 0) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) dzs[m] = dzl[m];
                    double r_loc = 0.0;
                    double dtl[4], dtu[4], dll[4], dlu[4], itl[4], itu[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double wv_ = NMPC_ZQ(z, q), dw = NMPC_ZQ(dzl, q);
                        const double rtl = mq[q] ? (wv_ - (lb[q] - cur[q]) - tl[q]) : 0.0;
                        const double rtu = mq[q] ? ((ub[q] - cur[q]) - wv_ - tu[q]) : 0.0;
                        itl[q] = w.rcp(tl[q]);
                        itu[q] = w.rcp(tu[q]);
                        dtl[q] = mq[q] ? (dw + rtl) : 0.0;
                        dtu[q] = mq[q] ? (-dw + rtu) : 0.0;
                        dll[q] = -ll[q] - ll[q] * itl[q] * dtl[q];
                        dlu[q] = -lu[q] - lu[q] * itu[q] * dtu[q];
                        r_loc = fmax(r_loc, fmax(-dtl[q] * itl[q], -dtu[q] * itu[q]));
                        r_loc = fmax(r_loc, mq[q] ? fmax(1.0 + dtl[q] * itl[q], 1.0 + dtu[q] * itu[q]) : 0.0);
                    }
                    const double a_aff = w.rcp(fmax(1.0, w.max(r_loc)));
                    double ma = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        ma += (tl[q] + a_aff * dtl[q]) * (ll[q] + a_aff * dll[q]) +
                              (tu[q] + a_aff * dtu[q]) * (lu[q] + a_aff * dlu[q]);
                    mu_aff = w.sum(ma) * inv_mcount;
                    const double sr = mu_aff * w.rcp(mu);
                    smu = sr * sr * sr * mu;
                    csc = fmin(1.0, 4.0 * a_aff);
                    if (act) {
                        double *sl = stg + lane * ST + O_D;
                        double dG[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double ctl = smu - csc * dtl[q] * dll[q], ctu = smu - csc * dtu[q] * dlu[q];
                            dG[q] = mq[q] ? (-ctl * itl[q] + ctu * itu[q]) : 0.0;
                        }
                        st2(sl, dG[0], dG[1]); st2(sl + 2, dG[2], dG[3]);
                        st2(sl + 4, 0.0, 0.0);
                    }
                } else {
                    double r_loc = 0.0;
                    double dtl[4], dtu[4], dll[4], dlu[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double wv_ = NMPC_ZQ(z, q), dwa = NMPC_ZQ(dzs, q);
                        const double rtl = mq[q] ? (wv_ - (lb[q] - cur[q]) - tl[q]) : 0.0;
                        const double rtu = mq[q] ? ((ub[q] - cur[q]) - wv_ - tu[q]) : 0.0;
                        const double itl = w.rcp(tl[q]), itu = w.rcp(tu[q]);
                        const double atl = mq[q] ? (dwa + rtl) : 0.0, atu = mq[q] ? (-dwa + rtu) : 0.0;
                        const double all_ = -ll[q] - ll[q] * itl * atl, alu = -lu[q] - lu[q] * itu * atu;
                        const double ctl = smu - csc * atl * all_, ctu = smu - csc * atu * alu;
                        const double dw = dwa + NMPC_ZQ(dzl, q);
                        dtl[q] = mq[q] ? (dw + rtl) : 0.0;
                        dtu[q] = mq[q] ? (-dw + rtu) : 0.0;
                        dll[q] = mq[q] ? (ctl * itl - ll[q] - ll[q] * itl * dtl[q]) : 0.0;
                        dlu[q] = mq[q] ? (ctu * itu - lu[q] - lu[q] * itu * dtu[q]) : 0.0;
                        r_loc = fmax(r_loc, fmax(-dtl[q] * itl, -dtu[q] * itu));
                        r_loc = fmax(r_loc, mq[q] ? fmax(-dll[q] * w.rcp(ll[q]), -dlu[q] * w.rcp(lu[q])) : 0.0);
                    }
                    const double a_max = w.rcp(w.max(r_loc));
                    const double tau = fmax(pb.tau, 1.0 - mu_aff);
                    const double alpha = fmin(1.0, tau * a_max);
#pragma unroll
                    for (int m = 0; m < 8; ++m) z[m] += alpha * (dzs[m] + dzl[m]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        tl[q] += alpha * dtl[q]; tu[q] += alpha * dtu[q];
                        ll[q] += alpha * dll[q]; lu[q] += alpha * dlu[q];
                    }
#pragma unroll
                    for (int m = 0; m < 6; ++m) rb[m] *= (1.0 - alpha);
                    lin_scale *= (1.0 - alpha);
                    ++it;
                }
            }
            w.stamp(pass == 0 ? 5 : 10);
            if (pass == 0) { w.sync(); w.stamp(6); }                      // barrier 3 (corrector rhs visible)
        }
    }
}

#undef NMPC_ZQ
#undef NMPC_VCOL
#undef NMPC_KOFF

}  // namespace nmpcq
