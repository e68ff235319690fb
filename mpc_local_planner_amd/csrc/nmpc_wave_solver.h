// nmpc_wave_solver.h -- one SQP-RTI solve of the path-tracking OCP by ONE 64-lane
// wavefront (gfx950).  This is the body of the HIP kernel in nmpc_hip.hip; it is
// written against a small "wave policy" W (lane id, LDS pointer, wave sync,
// cross-lane reads and reductions) so that tests can also run the very same
// source on a CPU lane emulator (tests/emu/) -- the product only ever
// instantiates it with the gfx950 policy.
//
// What it replaces (reference file:line):
//   Acados::Solver::solve                    src/AcadosSolver.cpp:28-76
//   -> setInitCondition / setParams          src/AcadosSolver.cpp:15-26, src/AcadosPathTracking.cpp:17-24
//   -> path_tracking_acados_solve (SQP-RTI)  src/AcadosPathTracking.cpp:26  [acados/HPIPM, not in tree]
//   model / cost / bounds                    acados/mpcPathTracking.py:41-47, 63-80, 100-115
//
// Work mapping inside the wave
//   * "stage layout": lane k owns shooting node k (x_k,u_k, slacks, multipliers).  Linearisation
//     (ERK4 + forward sensitivities, Gauss-Newton cost), all interior-point element-wise work and
//     step-length reductions run here, 40 of 64 lanes busy at N=40, no LDS traffic.
//   * "entry layout": lane (i,j) = (lane>>3, lane&7) owns entry (i,j) of the 8x8 stage matrices
//     in [u0,u1,X,Y,psi,v,delta,thr] order.  The Riccati factorisation sweep runs here, serial
//     in the stage index, three LDS exchanges per stage.
//   * the three vector-only sweeps per interior-point iteration run wave-uniform (every lane
//     carries the same 6-vector) straight from LDS broadcasts.
//   Per-stage matrices (dynamics Jacobian rows X,Y,psi; Riccati gains) live in LDS; rows v,delta,thr
//   of [B A] are compile/launch-time constants of this model and stay in registers.
#pragma once

#ifndef NMPC_DEV
#define NMPC_DEV inline
#endif

#include <math.h>

namespace nmpcw {

struct Problem {            // uniform over the launch
    int N;                  // horizon, N+1 <= 64
    double dt;              // shooting interval
    double inv_wheelbase;   // 1/L
    int ipm_max_iter;
    double mu0, thr0, tau, tol_mu, tol_lin, tol_comp;
    int sqp_max_iter;       // 1 = SQP-RTI (one QP per call)
    double sqp_tol;
};

struct InstanceIO {
    const double *x0;       // [6] measured state (lbx0 = ubx0, AcadosSolver.cpp:15-26)
    const double *p;        // [6] c0,c1,c2,c3,pitch,v_ref (AcadosPathTracking.h:31-34)
    const double *W;        // [7] diagonal NLS weights
    const double *bnd;      // [8] lbu0,lbu1,lbx4,lbx5, ubu0,ubu1,ubx4,ubx5
    double *traj;           // [(N+1)*8] iterate, per node X,Y,psi,v,delta,thr,u0,u1 (warm start, no shift)
    double *ctrl;           // [6] u0[2], x1.delta, x1.throttle, cost, (status + 16*ipm iterations)
    int *status;            // acados numbering: 0 ok, 1 NaN, 2 max iter, 4 QP failure
    int *iters;             // interior-point iterations (sum over SQP iterations)
    int *sqp_iters;         // QPs solved
    double *cost;
    int cold;               // != 0: start from x_k = x0, u_k = 0 (AcadosSolver.cpp:78-92)
};

// LDS layout, in doubles
constexpr int BA_STRIDE = 32;   // 8 columns x (3 variable rows + pad)
constexpr int SLOT = 28;
constexpr int S_K = 0;          // K[2][6]            (also: in  gam[4],Q00,Q01,Q02,0 | G[8])
constexpr int S_KFF = 12;       // kff[2]
constexpr int S_INV = 14;       // inv00,inv01,inv11
constexpr int S_D = 18;         // r_b[6] -> dz_aff[8] -> dG[4] -> dz_cor[8]
constexpr int S_TAB = 0;        // input view
constexpr int S_G = 8;
constexpr int X_P = 0, X_PV = 36, X_M = 44, X_V = 92, X_H = 100, X_XN = 120, X_SIZE = 128;
constexpr int lds_doubles(int N) { return N * (BA_STRIDE + SLOT) + X_SIZE; }

struct Lin {        // per-node linearisation kept in stage-layout registers
    double Q00, Q01, Q02;   // variable entries of the GN Hessian (X,X) (X,Y) (X,psi)
    double g[8];            // gradient [u0,u1,X,Y,psi,v,delta,thr]
    double rb[6];           // dynamics gap phi(x_k,u_k) - x_{k+1}
};


// sin/cos/tan for the bounded angles of this model (|x| < ~1e5): two-constant Cody-Waite
// reduction with fma and the classic degree-13/14 minimax kernels (errors < 1 ulp on the
// reduced argument).  Replaces the generic libm routines, whose huge-argument paths cost
// hundreds of VGPRs and scratch on gfx950 (hard part 3).
NMPC_DEV void sincos_bounded(double x, double *sn, double *cs)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma(fma(fma(fma(fma(1.58969099521155010221e-10, z, -2.50507602534068634195e-08), z,
                                          2.75573137070700676789e-06), z, -1.98412698298579493134e-04), z,
                                  8.33333333332248946124e-03), z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    const double pc = fma(fma(fma(fma(fma(-1.13596475577881948265e-11, z, 2.08757232129817482790e-09), z,
                                          -2.75573143513906633035e-07), z, 2.48015872894767294178e-05), z,
                                  -1.38888888888741095749e-03), z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = ((int)k) & 3;
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// ERK4 step with forward sensitivities of the kinematic bicycle, structure exploited:
//   rows delta,thr of phi are exact integrators, columns X,Y of d(phi)/dx are unit vectors,
//   the v-row is a constant-coefficient linear ODE.  sens columns: psi,v,delta,thr,u0,u1.
template <class W>
NMPC_DEV void rk4_sens(const Problem &pb, const double *x, const double *u, double gp,
                       double *phi, double *aX, double *aY, double *aP, double *aV)
{
    // Explicit sparsity: d(v_s) has columns (v,thr,u1) only; d(psi_s)/d(psi) == 1.
    const double h = pb.dt, iL = pb.inv_wheelbase;
    double kP = 0, kV = 0;
    double kPv = 0, kPd = 0, kPt = 0, kP0 = 0, kP1 = 0;   // d kP / d(v,delta,thr,u0,u1)
    double kVv = 0, kVt = 0, kV1 = 0;                     // d kV / d(v,thr,u1)
    double sX = 0, sY = 0, sP = 0, sV = 0;
    double sXp = 0, sXv = 0, sXd = 0, sXt = 0, sX0 = 0, sX1 = 0;
    double sYp = 0, sYv = 0, sYd = 0, sYt = 0, sY0 = 0, sY1 = 0;
    double sPv = 0, sPd = 0, sPt = 0, sP0 = 0, sP1 = 0;
    double sVv = 0, sVt = 0, sV1 = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double c = (s == 0) ? 0.0 : (s == 3 ? h : 0.5 * h);
        const double wgt = (s == 0 || s == 3) ? 1.0 : 2.0;
        const double ps = x[2] + c * kP, vs = x[3] + c * kV;
        const double ds = x[4] + c * u[0], ts = x[5] + c * u[1];
        // sensitivities of the stage point
        const double pv = c * kPv, pd = c * kPd, pt = c * kPt, p0 = c * kP0, p1 = c * kP1;  // d psi_s
        const double vv = 1.0 + c * kVv, vt = c * kVt, v1 = c * kV1;                        // d v_s
        double sp, cp, sd, cd;
        sincos_bounded(ps, &sp, &cp);
        sincos_bounded(ds, &sd, &cd);
        const double td = sd / cd;
        const double kX = vs * cp, kY = vs * sp;
        kP = vs * td * iL;
        kV = 5.0 * ts - 0.087 * vs + gp;
        const double tl = td * iL, vl = vs * (1.0 + td * td) * iL;
        // d kX = cp dv - kY dpsi ; d kY = sp dv + kX dpsi
        const double kXp = -kY, kXv = cp * vv - kY * pv, kXd = -kY * pd, kXt = cp * vt - kY * pt,
                     kX0 = -kY * p0, kX1 = cp * v1 - kY * p1;
        const double kYp = kX, kYv = sp * vv + kX * pv, kYd = kX * pd, kYt = sp * vt + kX * pt,
                     kY0 = kX * p0, kY1 = sp * v1 + kX * p1;
        // d kP = tl dv + vl d(delta_s),  d(delta_s) = e_delta + c e_u0
        kPv = tl * vv; kPd = vl; kPt = tl * vt; kP0 = vl * c; kP1 = tl * v1;
        // d kV = 5 d(thr_s) - 0.087 dv,  d(thr_s) = e_thr + c e_u1
        kVv = -0.087 * vv; kVt = 5.0 - 0.087 * vt; kV1 = 5.0 * c - 0.087 * v1;
        sX += wgt * kX; sY += wgt * kY; sP += wgt * kP; sV += wgt * kV;
        sXp += wgt * kXp; sXv += wgt * kXv; sXd += wgt * kXd; sXt += wgt * kXt; sX0 += wgt * kX0; sX1 += wgt * kX1;
        sYp += wgt * kYp; sYv += wgt * kYv; sYd += wgt * kYd; sYt += wgt * kYt; sY0 += wgt * kY0; sY1 += wgt * kY1;
        sPv += wgt * kPv; sPd += wgt * kPd; sPt += wgt * kPt; sP0 += wgt * kP0; sP1 += wgt * kP1;
        sVv += wgt * kVv; sVt += wgt * kVt; sV1 += wgt * kV1;
    }
    const double h6 = h / 6.0;
    phi[0] = x[0] + h6 * sX;
    phi[1] = x[1] + h6 * sY;
    phi[2] = x[2] + h6 * sP;
    phi[3] = x[3] + h6 * sV;
    phi[4] = x[4] + h * u[0];
    phi[5] = x[5] + h * u[1];
    // columns: psi, v, delta, thr, u0, u1
    aX[0] = h6 * sXp; aX[1] = h6 * sXv; aX[2] = h6 * sXd; aX[3] = h6 * sXt; aX[4] = h6 * sX0; aX[5] = h6 * sX1;
    aY[0] = h6 * sYp; aY[1] = h6 * sYv; aY[2] = h6 * sYd; aY[3] = h6 * sYt; aY[4] = h6 * sY0; aY[5] = h6 * sY1;
    aP[0] = 1.0;      aP[1] = h6 * sPv; aP[2] = h6 * sPd; aP[3] = h6 * sPt; aP[4] = h6 * sP0; aP[5] = h6 * sP1;
    aV[0] = 0.0;      aV[1] = 1.0 + h6 * sVv; aV[2] = 0.0; aV[3] = h6 * sVt; aV[4] = 0.0;     aV[5] = h6 * sV1;
}

template <class W>
NMPC_DEV void solve_instance(W &w, const Problem &pb, const InstanceIO &io)
{
    const int lane = w.lane();
    const int N = pb.N;
    const double h = pb.dt;
    double *lds = w.lds();
    double *ldsBA = lds;
    double *ldsSlot = lds + N * BA_STRIDE;
    double *ldsX = lds + N * (BA_STRIDE + SLOT);

    // ---------------------------------------------------------------- instance data (uniform)
    double x0[6], par[6], Wt[7], lb[4], ub[4];
#pragma unroll
    for (int m = 0; m < 6; ++m) { x0[m] = io.x0[m]; par[m] = io.p[m]; }
#pragma unroll
    for (int m = 0; m < 7; ++m) Wt[m] = io.W[m];
#pragma unroll
    for (int m = 0; m < 4; ++m) { lb[m] = io.bnd[m]; ub[m] = io.bnd[4 + m]; }
    double gp, gpc;
    sincos_bounded(par[4], &gp, &gpc);
    gp *= 9.81;

    // ---------------------------------------------------------------- iterate (stage layout)
    const bool has_node = lane <= N;       // lanes 0..N hold x_k
    const bool act = lane < N;             // lanes 0..N-1 hold u_k and a shooting interval
    const int node = has_node ? lane : N;
    double xb[6], ub_[2];                   // NLP iterate of this node
    {
        const double *t = io.traj + node * 8;
#pragma unroll
        for (int m = 0; m < 6; ++m) xb[m] = io.cold ? x0[m] : t[m];
        ub_[0] = (io.cold || !act) ? 0.0 : t[6];
        ub_[1] = (io.cold || !act) ? 0.0 : t[7];
    }
    const bool actx = act && lane >= 1;    // state bounds exist on nodes 1..N-1 only
    const bool mq[4] = {act, act, actx, actx};
    const double mcount = (double)(4 * N + 4 * (N - 1));

    // entry-layout constants
    const int ei = lane >> 3, ej = lane & 7;
    int total_ipm = 0, status = 0, sqp_it = 0;
    double cost_out = 0.0;
    bool bot_ready = false;
    double cjb[3] = {0, 0, 0}, cib[3] = {0, 0, 0};   // rows v,delta,thr of columns ej / ei of [B A]
    double a33 = 0, a35 = 0, b31 = 0;

    // constant part of the stage Hessian seen by entry (ei,ej) and where its variable part sits
    double hconst = 0.0;
    int htab = 7;
    if (ei == ej) {
        if (ei == 0) { hconst = h * Wt[5]; htab = 0; }
        if (ei == 1) { hconst = h * Wt[6]; htab = 1; }
        if (ei == 2) { htab = 4; }
        if (ei == 3) hconst = h * Wt[0];
        if (ei == 4) hconst = h * Wt[1];
        if (ei == 5) hconst = h * Wt[2];
        if (ei == 6) { hconst = h * Wt[3]; htab = 2; }
        if (ei == 7) { hconst = h * Wt[4]; htab = 3; }
    } else if ((ei == 2 && ej == 3) || (ei == 3 && ej == 2)) {
        htab = 5;
    } else if ((ei == 2 && ej == 4) || (ei == 4 && ej == 2)) {
        htab = 6;
    }
    const int prow = (ei >= 2) ? (ei - 2) : (ej < 6 ? ej : 5);   // row of P this lane multiplies with

    for (sqp_it = 0; sqp_it < pb.sqp_max_iter; ++sqp_it) {
        // ------------------------------------------------------------ linearise (stage layout)
        Lin L;
        double dx0[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) dx0[m] = x0[m] - w.readlane(xb[m], 0);
        {
            double phi[6], aX[6], aY[6], aP[6], aV[6];
            rk4_sens<W>(pb, xb, ub_, gp, phi, aX, aY, aP, aV);
            // x_{k+1} from the next lane
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const double xn = w.shfl_down1(xb[m]);
                L.rb[m] = act ? (phi[m] - xn) : 0.0;
            }
            if (lane == 0) {   // fold the known dx_0 into the first gap: b_0 += A_0 dx0
                L.rb[0] += dx0[0] + aX[0] * dx0[2] + aX[1] * dx0[3] + aX[2] * dx0[4] + aX[3] * dx0[5];
                L.rb[1] += dx0[1] + aY[0] * dx0[2] + aY[1] * dx0[3] + aY[2] * dx0[4] + aY[3] * dx0[5];
                L.rb[2] += aP[0] * dx0[2] + aP[1] * dx0[3] + aP[2] * dx0[4] + aP[3] * dx0[5];
                L.rb[3] += aV[1] * dx0[3] + aV[3] * dx0[5];
                L.rb[4] += dx0[4];
                L.rb[5] += dx0[5];
            }
            // variable rows X,Y,psi of [B A] -> LDS, column-major, column order u0,u1,X,Y,psi,v,delta,thr
            if (act) {
                double *c = ldsBA + lane * BA_STRIDE;
                c[0] = aX[4];  c[1] = aY[4];  c[2] = aP[4];  c[3] = 0.0;   // u0
                c[4] = aX[5];  c[5] = aY[5];  c[6] = aP[5];  c[7] = 0.0;   // u1
                c[8] = 1.0;    c[9] = 0.0;    c[10] = 0.0;   c[11] = 0.0;  // X
                c[12] = 0.0;   c[13] = 1.0;   c[14] = 0.0;   c[15] = 0.0;  // Y
                c[16] = aX[0]; c[17] = aY[0]; c[18] = aP[0]; c[19] = 0.0;  // psi
                c[20] = aX[1]; c[21] = aY[1]; c[22] = aP[1]; c[23] = 0.0;  // v
                c[24] = aX[2]; c[25] = aY[2]; c[26] = aP[2]; c[27] = 0.0;  // delta
                c[28] = aX[3]; c[29] = aY[3]; c[30] = aP[3]; c[31] = 0.0;  // thr
            }
            if (!bot_ready) {   // constant rows (v,delta,thr) of [B A]: identical on every lane
                a33 = w.readlane(aV[1], 0);
                a35 = w.readlane(aV[3], 0);
                b31 = w.readlane(aV[5], 0);
                const double bot[8][3] = {{0, h, 0}, {b31, 0, h}, {0, 0, 0}, {0, 0, 0},
                                          {0, 0, 0}, {a33, 0, 0}, {0, 1, 0}, {a35, 0, 1}};
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (ej == c) cjb[r] = bot[c][r];
                        if (ei == c) cib[r] = bot[c][r];
                    }
                bot_ready = true;
            }
            // Gauss-Newton cost terms (mpcPathTracking.py:63-80,103)
            const double X = xb[0];
            const double s = 3.0 * par[3] * X * X + 2.0 * par[2] * X + par[1];
            const double dsl = 6.0 * par[3] * X + 2.0 * par[2];
            const double ypath = par[3] * X * X * X + par[2] * X * X + par[1] * X + par[0];
            const double cte = ypath - xb[1];
            const double epsi = xb[2] - atan(s);
            const double d = -dsl / (1.0 + s * s);
            const double ev = xb[3] - par[5];
            L.Q00 = h * (Wt[0] * s * s + Wt[1] * d * d);
            L.Q01 = -h * Wt[0] * s;
            L.Q02 = h * Wt[1] * d;
            L.g[0] = h * Wt[5] * ub_[0];
            L.g[1] = h * Wt[6] * ub_[1];
            L.g[2] = h * (Wt[0] * s * cte + Wt[1] * d * epsi);
            L.g[3] = -h * Wt[0] * cte;
            L.g[4] = h * Wt[1] * epsi;
            L.g[5] = h * Wt[2] * ev;
            L.g[6] = h * Wt[3] * xb[4];
            L.g[7] = h * Wt[4] * xb[5];
            const double cst = 0.5 * h * (Wt[0] * cte * cte + Wt[1] * epsi * epsi + Wt[2] * ev * ev +
                                          Wt[3] * xb[4] * xb[4] + Wt[4] * xb[5] * xb[5] +
                                          Wt[5] * ub_[0] * ub_[0] + Wt[6] * ub_[1] * ub_[1]);
            cost_out = w.sum(act ? cst : 0.0);
        }
        const double R00 = h * Wt[5], R11 = h * Wt[6], Q11 = h * Wt[0], Q22 = h * Wt[1],
                     Q33 = h * Wt[2], Q44 = h * Wt[3], Q55 = h * Wt[4];

        // ------------------------------------------------------------ interior point (stage layout)
        // bounded variables of this node: u0,u1,delta,thr -> z index 0,1,6,7
        double tl[4], tu[4], ll[4], lu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double cur = (q == 0) ? ub_[0] : (q == 1) ? ub_[1] : (q == 2) ? xb[4] : xb[5];
            const double a = fmax(cur - lb[q], pb.thr0), b = fmax(ub[q] - cur, pb.thr0);
            tl[q] = mq[q] ? a : 1.0;
            tu[q] = mq[q] ? b : 1.0;
            ll[q] = mq[q] ? pb.mu0 * w.rcp(a) : 0.0;
            lu[q] = mq[q] ? pb.mu0 * w.rcp(b) : 0.0;
        }
        double z[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // QP iterate of this node [du0,du1,dX..dthr]
        double lin_scale = 1.0;
        int it = 0, qp_status = 4;
        w.sync();   // BA columns visible
        for (;; ++it) {
            // ---- E-A: barrier system of this node
            double mu_loc = 0.0, c_loc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mu_loc += tl[q] * ll[q] + tu[q] * lu[q];
                // pairwise complementarity (masked pairs hold t = 1, lambda = 0)
                c_loc = fmax(c_loc, fmax(fmin(tl[q], ll[q]), fmin(tu[q], lu[q])));
            }
            const double mu = w.sum(mu_loc) / mcount;
            const double cmax = w.max(c_loc);
            if (!(mu == mu)) { qp_status = 1; break; }
            if (it > 0 && mu <= pb.tol_mu && lin_scale <= pb.tol_lin && cmax <= pb.tol_comp) { qp_status = 0; break; }
            if (it >= pb.ipm_max_iter) { qp_status = (mu <= 1e-8 && lin_scale <= 1e-8) ? 0 : 4; break; }

            double G[8];
            G[0] = R00 * z[0] + L.g[0];
            G[1] = R11 * z[1] + L.g[1];
            G[2] = L.Q00 * z[2] + L.Q01 * z[3] + L.Q02 * z[4] + L.g[2];
            G[3] = L.Q01 * z[2] + Q11 * z[3] + L.g[3];
            G[4] = L.Q02 * z[2] + Q22 * z[4] + L.g[4];
            G[5] = Q33 * z[5] + L.g[5];
            G[6] = Q44 * z[6] + L.g[6];
            G[7] = Q55 * z[7] + L.g[7];
            double gam[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double cur = (q == 0) ? ub_[0] : (q == 1) ? ub_[1] : (q == 2) ? xb[4] : xb[5];
                const double wv = (q == 0) ? z[0] : (q == 1) ? z[1] : (q == 2) ? z[6] : z[7];
                const double rtl = mq[q] ? (wv - (lb[q] - cur) - tl[q]) : 0.0;
                const double rtu = mq[q] ? ((ub[q] - cur) - wv - tu[q]) : 0.0;
                const double rl = ll[q] * w.rcp(tl[q]), ru = lu[q] * w.rcp(tu[q]);
                gam[q] = rl + ru;
                const double gq = rl * rtl - ru * rtu;
                if (q == 0) G[0] += gq;
                if (q == 1) G[1] += gq;
                if (q == 2) G[6] += gq;
                if (q == 3) G[7] += gq;
            }
            if (act) {
                double *sl = ldsSlot + lane * SLOT;
                sl[S_TAB + 0] = gam[0]; sl[S_TAB + 1] = gam[1]; sl[S_TAB + 2] = gam[2]; sl[S_TAB + 3] = gam[3];
                sl[S_TAB + 4] = L.Q00;  sl[S_TAB + 5] = L.Q01;  sl[S_TAB + 6] = L.Q02;  sl[S_TAB + 7] = 0.0;
#pragma unroll
                for (int m = 0; m < 8; ++m) sl[S_G + m] = G[m];
#pragma unroll
                for (int m = 0; m < 6; ++m) sl[S_D + m] = L.rb[m];
            }
            // terminal cost-to-go: none (no *_e cost or bound in mpcPathTracking.py)
            if (lane < 36) ldsX[X_P + lane] = 0.0;
            if (lane < 8) ldsX[X_PV + lane] = 0.0;
            w.sync();

            // ---- sweep B: Riccati factorisation, entry layout, k = N-1 .. 0
#pragma unroll 1
            for (int k = N - 1; k >= 0; --k) {
                const double *ba = ldsBA + k * BA_STRIDE;
                double *sl = ldsSlot + k * SLOT;
                double cj[6], ci[6], pr[6], rbk[6];
#pragma unroll
                for (int m = 0; m < 3; ++m) { cj[m] = ba[ej * 4 + m]; ci[m] = ba[ei * 4 + m]; }
#pragma unroll
                for (int m = 0; m < 3; ++m) { cj[3 + m] = cjb[m]; ci[3 + m] = cib[m]; }
#pragma unroll
                for (int m = 0; m < 6; ++m) { pr[m] = ldsX[X_P + prow * 6 + m]; rbk[m] = sl[S_D + m]; }
                const double hc = hconst + sl[S_TAB + htab];
                const double Gj = sl[S_G + ej];
                double Mv = 0.0, vv = ldsX[X_PV + prow];
#pragma unroll
                for (int m = 0; m < 6; ++m) { Mv += pr[m] * cj[m]; vv += pr[m] * rbk[m]; }
                if (ei >= 2) ldsX[X_M + ej * 6 + (ei - 2)] = Mv;        // M = P [B A], stored by column
                if (ei == 0 && ej < 6) ldsX[X_V + ej] = vv;             // v = P r_b + p
                w.sync();
                double Hv = hc, hj = Gj;
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    Hv += ci[m] * ldsX[X_M + ej * 6 + m];               // H = Hcost + [B A]' M
                    hj += cj[m] * ldsX[X_V + m];                        // h = G + [B A]' v
                }
                if (ei < 2) ldsX[X_H + ej * 2 + ei] = Hv;               // rows u0,u1 of H
                const double H00 = w.readlane(Hv, 0), H01 = w.readlane(Hv, 1), H11 = w.readlane(Hv, 9);
                const double h0 = w.readlane(hj, 0), h1 = w.readlane(hj, 1);
                const double rdet = w.rcp(H00 * H11 - H01 * H01);
                const double i00 = H11 * rdet, i01 = -H01 * rdet, i11 = H00 * rdet;
                w.sync();
                const double Hj0 = ldsX[X_H + ej * 2], Hj1 = ldsX[X_H + ej * 2 + 1];
                const double Hi0 = ldsX[X_H + ei * 2], Hi1 = ldsX[X_H + ei * 2 + 1];
                const double K0 = -(i00 * Hj0 + i01 * Hj1), K1 = -(i01 * Hj0 + i11 * Hj1);
                const double Pn = Hv + Hi0 * K0 + Hi1 * K1;
                const double pn = hj + K0 * h0 + K1 * h1;
                if (ei >= 2 && ej >= ei) {                              // upper triangle, mirrored
                    ldsX[X_P + (ei - 2) * 6 + (ej - 2)] = Pn;
                    ldsX[X_P + (ej - 2) * 6 + (ei - 2)] = Pn;
                }
                if (ei == 2 && ej >= 2) ldsX[X_PV + (ej - 2)] = pn;
                if (ei < 2 && ej >= 2) sl[S_K + ei * 6 + (ej - 2)] = (ei == 0) ? K0 : K1;
                if (lane == 0) {
                    sl[S_KFF + 0] = -(i00 * h0 + i01 * h1);
                    sl[S_KFF + 1] = -(i01 * h0 + i11 * h1);
                    sl[S_INV + 0] = i00; sl[S_INV + 1] = i01; sl[S_INV + 2] = i11;
                }
                w.sync();
            }

            // ---- sweeps F1 / Bv / F2: wave-uniform vector recursions
            double dzs[8];       // affine step of this node (stage layout)
            double ctl[4], ctu[4], smu = 0.0, mu_aff = 0.0;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {
                    // backward vector sweep with the corrector right-hand side
                    double p6[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
                    for (int k = N - 1; k >= 0; --k) {
                        const double *ba = ldsBA + k * BA_STRIDE;
                        double *sl = ldsSlot + k * SLOT;
                        const double dg0 = sl[S_D + 0], dg1 = sl[S_D + 1], dg2 = sl[S_D + 2], dg3 = sl[S_D + 3];
                        const double hu0 = dg0 + ba[0] * p6[0] + ba[1] * p6[1] + ba[2] * p6[2] + h * p6[4];
                        const double hu1 = dg1 + ba[4] * p6[0] + ba[5] * p6[1] + ba[6] * p6[2] + b31 * p6[3] + h * p6[5];
                        double hx[6];
                        hx[0] = p6[0];
                        hx[1] = p6[1];
                        hx[2] = p6[2] + ba[16] * p6[0] + ba[17] * p6[1];
                        hx[3] = a33 * p6[3] + ba[20] * p6[0] + ba[21] * p6[1] + ba[22] * p6[2];
                        hx[4] = dg2 + p6[4] + ba[24] * p6[0] + ba[25] * p6[1] + ba[26] * p6[2];
                        hx[5] = dg3 + p6[5] + ba[28] * p6[0] + ba[29] * p6[1] + ba[30] * p6[2] + a35 * p6[3];
                        const double i00 = sl[S_INV + 0], i01 = sl[S_INV + 1], i11 = sl[S_INV + 2];
#pragma unroll
                        for (int c = 0; c < 6; ++c) p6[c] = hx[c] + sl[S_K + c] * hu0 + sl[S_K + 6 + c] * hu1;
                        if (lane == 0) {
                            sl[S_KFF + 0] = -(i00 * hu0 + i01 * hu1);
                            sl[S_KFF + 1] = -(i01 * hu0 + i11 * hu1);
                        }
                    }
                    w.sync();
                }
                // forward sweep
                double dx[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
                for (int k = 0; k < N; ++k) {
                    const double *ba = ldsBA + k * BA_STRIDE;
                    double *sl = ldsSlot + k * SLOT;
                    double rbk[6];
#pragma unroll
                    for (int m = 0; m < 6; ++m) rbk[m] = (pass == 0) ? sl[S_D + m] : 0.0;
                    double du0 = sl[S_KFF + 0], du1 = sl[S_KFF + 1];
#pragma unroll
                    for (int c = 0; c < 6; ++c) { du0 += sl[S_K + c] * dx[c]; du1 += sl[S_K + 6 + c] * dx[c]; }
                    double dn[6];
                    dn[0] = rbk[0] + dx[0] + ba[16] * dx[2] + ba[20] * dx[3] + ba[24] * dx[4] + ba[28] * dx[5] + ba[0] * du0 + ba[4] * du1;
                    dn[1] = rbk[1] + dx[1] + ba[17] * dx[2] + ba[21] * dx[3] + ba[25] * dx[4] + ba[29] * dx[5] + ba[1] * du0 + ba[5] * du1;
                    dn[2] = rbk[2] + dx[2] + ba[22] * dx[3] + ba[26] * dx[4] + ba[30] * dx[5] + ba[2] * du0 + ba[6] * du1;
                    dn[3] = rbk[3] + a33 * dx[3] + a35 * dx[5] + b31 * du1;
                    dn[4] = rbk[4] + dx[4] + h * du0;
                    dn[5] = rbk[5] + dx[5] + h * du1;
                    w.sync_lockstep();   // every lane has read r_b of this slot (free on a real wave)
                    if (lane == 0) {
                        sl[S_D + 0] = du0; sl[S_D + 1] = du1;
#pragma unroll
                        for (int m = 0; m < 6; ++m) sl[S_D + 2 + m] = dx[m];
                    }
#pragma unroll
                    for (int m = 0; m < 6; ++m) dx[m] = dn[m];
                }
                if (lane == 0) {
                    ldsX[X_XN + 0] = 0.0; ldsX[X_XN + 1] = 0.0;
#pragma unroll
                    for (int m = 0; m < 6; ++m) ldsX[X_XN + 2 + m] = dx[m];
                }
                w.sync();
                // ---- stage layout: pick up the direction of this node
                double dzl[8];
                {
                    const double *src = act ? (ldsSlot + lane * SLOT + S_D) : (ldsX + X_XN);
#pragma unroll
                    for (int m = 0; m < 8; ++m) dzl[m] = has_node ? src[m] : 0.0;
                }
                // residuals and ratios of this node at the current (unchanged) iterate
                double rtl[4], rtu[4], itl[4], itu[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double cur = (q == 0) ? ub_[0] : (q == 1) ? ub_[1] : (q == 2) ? xb[4] : xb[5];
                    const double wv = (q == 0) ? z[0] : (q == 1) ? z[1] : (q == 2) ? z[6] : z[7];
                    rtl[q] = mq[q] ? (wv - (lb[q] - cur) - tl[q]) : 0.0;
                    rtu[q] = mq[q] ? ((ub[q] - cur) - wv - tu[q]) : 0.0;
                    itl[q] = w.rcp(tl[q]);
                    itu[q] = w.rcp(tu[q]);
                }
                if (pass == 0) {
                    // E-B: affine step lengths, centring parameter, corrector right-hand side
#pragma unroll
                    for (int m = 0; m < 8; ++m) dzs[m] = dzl[m];
                    double r_loc = 0.0;
                    double dtl[4], dtu[4], dll[4], dlu[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double dw = (q == 0) ? dzl[0] : (q == 1) ? dzl[1] : (q == 2) ? dzl[6] : dzl[7];
                        dtl[q] = mq[q] ? (dw + rtl[q]) : 0.0;
                        dtu[q] = mq[q] ? (-dw + rtu[q]) : 0.0;
                        dll[q] = -ll[q] - ll[q] * itl[q] * dtl[q];
                        dlu[q] = -lu[q] - lu[q] * itu[q] * dtu[q];
                        // largest -d/v over the positive variables (step to the boundary = 1/that)
                        r_loc = fmax(r_loc, fmax(-dtl[q] * itl[q], -dtu[q] * itu[q]));
                        r_loc = fmax(r_loc, mq[q] ? fmax(1.0 + dtl[q] * itl[q], 1.0 + dtu[q] * itu[q]) : 0.0);
                    }
                    const double a_aff = w.rcp(fmax(1.0, w.max(r_loc)));
                    double ma = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        ma += (tl[q] + a_aff * dtl[q]) * (ll[q] + a_aff * dll[q]) +
                              (tu[q] + a_aff * dtu[q]) * (lu[q] + a_aff * dlu[q]);
                    mu_aff = w.sum(ma) / mcount;
                    const double sr = mu_aff / mu;
                    smu = sr * sr * sr * mu;
                    // Mehrotra's second-order term, damped when the affine step is almost fully
                    // blocked (undamped it over-corrects and the iteration can cycle)
                    const double csc = fmin(1.0, 4.0 * a_aff);
                    double dG[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ctl[q] = smu - csc * dtl[q] * dll[q];
                        ctu[q] = smu - csc * dtu[q] * dlu[q];
                        dG[q] = mq[q] ? (-ctl[q] * itl[q] + ctu[q] * itu[q]) : 0.0;
                    }
                    if (act) {
                        double *sl = ldsSlot + lane * SLOT;
#pragma unroll
                        for (int q = 0; q < 4; ++q) sl[S_D + q] = dG[q];
                    }
                    w.sync();
                } else {
                    // E-C: combined direction, step length, update
#pragma unroll
                    for (int m = 0; m < 8; ++m) dzs[m] += dzl[m];
                    double r_loc = 0.0;
                    double dtl[4], dtu[4], dll[4], dlu[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double dw = (q == 0) ? dzs[0] : (q == 1) ? dzs[1] : (q == 2) ? dzs[6] : dzs[7];
                        dtl[q] = mq[q] ? (dw + rtl[q]) : 0.0;
                        dtu[q] = mq[q] ? (-dw + rtu[q]) : 0.0;
                        dll[q] = mq[q] ? (ctl[q] * itl[q] - ll[q] - ll[q] * itl[q] * dtl[q]) : 0.0;
                        dlu[q] = mq[q] ? (ctu[q] * itu[q] - lu[q] - lu[q] * itu[q] * dtu[q]) : 0.0;
                        r_loc = fmax(r_loc, fmax(-dtl[q] * itl[q], -dtu[q] * itu[q]));
                        r_loc = fmax(r_loc, mq[q] ? fmax(-dll[q] * w.rcp(ll[q]), -dlu[q] * w.rcp(lu[q])) : 0.0);
                    }
                    const double a_max = w.rcp(w.max(r_loc));
                    const double tau = fmax(pb.tau, 1.0 - mu_aff);
                    const double alpha = fmin(1.0, tau * a_max);
#pragma unroll
                    for (int m = 0; m < 8; ++m) z[m] += alpha * dzs[m];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        tl[q] += alpha * dtl[q]; tu[q] += alpha * dtu[q];
                        ll[q] += alpha * dll[q]; lu[q] += alpha * dlu[q];
                    }
#pragma unroll
                    for (int m = 0; m < 6; ++m) L.rb[m] *= (1.0 - alpha);
                    lin_scale *= (1.0 - alpha);
                }
            }
        }
        total_ipm += it;

        // ------------------------------------------------------------ full step (alpha = 1, no line search)
        bool bad = false;
#pragma unroll
        for (int m = 0; m < 8; ++m) bad = bad || !(fabs(z[m]) <= 1e300);
        const double nbad = w.sum((bad && has_node) ? 1.0 : 0.0);
        if (qp_status == 0 && nbad != 0.0) qp_status = 1;
        if (qp_status == 1) { status = 1; ++sqp_it; break; }
        ub_[0] += z[0];
        ub_[1] += z[1];
#pragma unroll
        for (int m = 0; m < 6; ++m) xb[m] += z[2 + m];
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < 6; ++m) xb[m] = x0[m];
        }
        status = qp_status;
        if (qp_status != 0) { ++sqp_it; break; }
        if (pb.sqp_max_iter > 1) {
            double sl = 0.0;
#pragma unroll
            for (int m = 0; m < 8; ++m) sl = fmax(sl, fabs(z[m]));
            const double step = w.max(has_node ? sl : 0.0);
            if (step <= pb.sqp_tol) { ++sqp_it; status = 0; break; }
            if (sqp_it + 1 >= pb.sqp_max_iter) status = 2;
        }
    }

    // ---------------------------------------------------------------- write back
    if (has_node) {
        double *t = io.traj + lane * 8;
#pragma unroll
        for (int m = 0; m < 6; ++m) t[m] = xb[m];
        t[6] = act ? ub_[0] : 0.0;
        t[7] = act ? ub_[1] : 0.0;
    }
    const double u00 = w.readlane(ub_[0], 0), u01 = w.readlane(ub_[1], 0);
    const double d1 = w.readlane(xb[4], 1), t1 = w.readlane(xb[5], 1);
    if (lane == 0) {
        io.ctrl[0] = u00; io.ctrl[1] = u01; io.ctrl[2] = d1; io.ctrl[3] = t1;
        io.ctrl[4] = cost_out;
        io.ctrl[5] = (double)(status + 16 * total_ipm);
        *io.status = status;
        *io.iters = total_ipm;
        *io.sqp_iters = sqp_it;
        *io.cost = cost_out;
    }
}

}  // namespace nmpcw
