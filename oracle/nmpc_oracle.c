/*
 * nmpc_oracle.c -- CPU restatement of the reference hot path (see nmpc_oracle.h
 * for scope, citations and parity status: acados parity unpinned).
 * TEST INFRASTRUCTURE ONLY.  Serial, dense, deliberately unoptimised: every
 * matrix is handled as a general dense matrix so that the structure-exploiting
 * HIP kernels are checked against an independent formulation.
 */
#include "nmpc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NX ORC_NX
#define NU ORC_NU
#define NZ ORC_NZ
#define NB 4 /* bounded variables per stage: u0,u1,x4,x5 */

/* z-index (ordering [x;u]) of bounded variable j */
static const int BIDX[NB] = {6, 7, 4, 5};

void orc_default_config(orc_config *c, int N, double dt)
{
    static const double W[ORC_NY] = {5, 5, 10, 0.01, 0.1, 0.5, 0.1}; /* mpcPathTracking.py:103 */
    memset(c, 0, sizeof(*c));
    c->N = N;
    c->dt = dt;
    c->wheelbase = 2.65;                 /* mpc_params.yaml:59 */
    memcpy(c->W, W, sizeof(W));
    c->lbu[0] = -80.0;  c->ubu[0] = 80.0;  /* steering_angle_dot_range :56 */
    c->lbu[1] = -0.33;  c->ubu[1] = 0.33;  /* throttle_dot_range :50 */
    c->lbx[0] = -0.57;  c->ubx[0] = 0.57;  /* steering_angle_range :53 */
    c->lbx[1] = -1.0;   c->ubx[1] = 1.0;   /* throttle_range :47 */
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
}

/* ---------------------------------------------------------------- model */

void orc_f(const orc_config *c, const double *x, const double *u, const double *p, double *f)
{
    const double psi = x[2], v = x[3], delta = x[4], thr = x[5];
    f[0] = v * cos(psi);
    f[1] = v * sin(psi);
    f[2] = v / c->wheelbase * tan(delta);
    f[3] = 5.0 * thr - 0.087 * v + sin(p[4]) * 9.81;
    f[4] = u[0];
    f[5] = u[1];
}

/* df/dx (6x6 row-major); df/du is rows 4,5 = identity */
static void f_jac_x(const orc_config *c, const double *x, double *fx)
{
    const double psi = x[2], v = x[3], delta = x[4];
    const double cd = cos(delta);
    memset(fx, 0, sizeof(double) * NX * NX);
    fx[0 * NX + 2] = -v * sin(psi);
    fx[0 * NX + 3] = cos(psi);
    fx[1 * NX + 2] = v * cos(psi);
    fx[1 * NX + 3] = sin(psi);
    fx[2 * NX + 3] = tan(delta) / c->wheelbase;
    fx[2 * NX + 4] = v / (c->wheelbase * cd * cd);
    fx[3 * NX + 3] = -0.087;
    fx[3 * NX + 5] = 5.0;
}

void orc_rk4_sens(const orc_config *c, const double *x, const double *u, const double *p,
                  double *phi, double *A, double *B)
{
    static const double a[4] = {0.0, 0.5, 0.5, 1.0};
    static const double w[4] = {1.0, 2.0, 2.0, 1.0};
    double K[4][NX], dK[4][NX * NZ]; /* sensitivities wrt [x(6) | u(2)] */
    double xs[NX], S[NX * NZ], fx[NX * NX];
    const double h = c->dt;
    for (int s = 0; s < 4; ++s) {
        for (int i = 0; i < NX; ++i) {
            xs[i] = x[i] + (s ? a[s] * h * K[s - 1][i] : 0.0);
            for (int j = 0; j < NZ; ++j)
                S[i * NZ + j] = (i == j ? 1.0 : 0.0) + (s ? a[s] * h * dK[s - 1][i * NZ + j] : 0.0);
        }
        orc_f(c, xs, u, p, K[s]);
        f_jac_x(c, xs, fx);
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NZ; ++j) {
                double acc = 0.0;
                for (int l = 0; l < NX; ++l) acc += fx[i * NX + l] * S[l * NZ + j];
                dK[s][i * NZ + j] = acc;
            }
        dK[s][4 * NZ + 6] += 1.0;
        dK[s][5 * NZ + 7] += 1.0;
    }
    for (int i = 0; i < NX; ++i) {
        double acc = 0.0;
        for (int s = 0; s < 4; ++s) acc += w[s] * K[s][i];
        phi[i] = x[i] + h / 6.0 * acc;
        for (int j = 0; j < NZ; ++j) {
            double d = 0.0;
            for (int s = 0; s < 4; ++s) d += w[s] * dK[s][i * NZ + j];
            d = (i == j ? 1.0 : 0.0) + h / 6.0 * d;
            if (j < NX) A[i * NX + j] = d; else B[i * NU + (j - NX)] = d;
        }
    }
}

/* ---------------------------------------------------------------- cost */

void orc_cost_yJ(const orc_config *c, const double *x, const double *u, const double *p,
                 double *y, double *J)
{
    (void)c;
    const double x1 = x[0], y1 = x[1], psi = x[2], v = x[3];
    const double c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3], vref = p[5];
    const double s = 3.0 * c3 * x1 * x1 + 2.0 * c2 * x1 + c1;   /* path slope */
    const double ds = 6.0 * c3 * x1 + 2.0 * c2;
    const double ypath = c3 * x1 * x1 * x1 + c2 * x1 * x1 + c1 * x1 + c0;
    y[0] = ypath - y1;
    y[1] = psi - atan(s);
    y[2] = v - vref;
    y[3] = x[4];
    y[4] = x[5];
    y[5] = u[0];
    y[6] = u[1];
    memset(J, 0, sizeof(double) * ORC_NY * NZ);
    J[0 * NZ + 0] = s;
    J[0 * NZ + 1] = -1.0;
    J[1 * NZ + 0] = -ds / (1.0 + s * s);
    J[1 * NZ + 2] = 1.0;
    J[2 * NZ + 3] = 1.0;
    J[3 * NZ + 4] = 1.0;
    J[4 * NZ + 5] = 1.0;
    J[5 * NZ + 6] = 1.0;
    J[6 * NZ + 7] = 1.0;
}

void orc_stage_gn(const orc_config *c, const double *x, const double *u, const double *p,
                  double *H, double *g)
{
    double y[ORC_NY], J[ORC_NY * NZ];
    orc_cost_yJ(c, x, u, p, y, J);
    for (int i = 0; i < NZ; ++i) {
        double gi = 0.0;
        for (int r = 0; r < ORC_NY; ++r) gi += J[r * NZ + i] * c->W[r] * y[r];
        g[i] = c->dt * gi;
        for (int j = 0; j < NZ; ++j) {
            double hij = 0.0;
            for (int r = 0; r < ORC_NY; ++r) hij += J[r * NZ + i] * c->W[r] * J[r * NZ + j];
            H[i * NZ + j] = c->dt * hij;
        }
    }
}

static double stage_cost(const orc_config *c, const double *x, const double *u, const double *p)
{
    double y[ORC_NY], J[ORC_NY * NZ], acc = 0.0;
    orc_cost_yJ(c, x, u, p, y, J);
    for (int r = 0; r < ORC_NY; ++r) acc += c->W[r] * y[r] * y[r];
    return 0.5 * c->dt * acc;
}

/* ---------------------------------------------------------------- QP: Riccati IPM */

typedef struct ric_ws {
    int N;
    double *K;    /* N*2*6 feedback gains, du = K dx + kff */
    double *Hinv; /* N*4  inverse of Huu */
    double *kff;  /* N*2 */
} ric_ws;

/* 2x2 SPD inverse */
static void inv2(const double *M, double *Mi)
{
    const double det = M[0] * M[3] - M[1] * M[2];
    const double id = 1.0 / det;
    Mi[0] = M[3] * id;  Mi[1] = -M[1] * id;
    Mi[2] = -M[2] * id; Mi[3] = M[0] * id;
}

/* Backward Riccati sweep.  Hd = Hessian blocks with barrier diagonal already
 * added (N*64, [x;u]), G = gradient (N*8), rb = dynamics residual (N*6).
 * factor != 0: compute and store K, Hinv; factor == 0: reuse them, rb ignored
 * (homogeneous dynamics).  Forward sweep follows: dz[(N+1)*8], dx_0 = 0. */
static void riccati_solve(const orc_qp *qp, ric_ws *w, const double *Hd, const double *G,
                          const double *rb, int factor, double *dz)
{
    const int N = qp->N;
    double P[NX * NX], p[NX];
    memset(P, 0, sizeof(P));  /* no terminal cost (mpcPathTracking.py sets no *_e field) */
    memset(p, 0, sizeof(p));
    for (int k = N - 1; k >= 0; --k) {
        const double *A = qp->A + k * 36, *B = qp->B + k * 12;
        const double *H = Hd + k * 64, *Gk = G + k * 8;
        double *K = w->K + k * 12, *Hi = w->Hinv + k * 4, *kf = w->kff + k * 2;
        double v[NX], hu[NU], hx[NX];
        /* v = P rb + p */
        for (int i = 0; i < NX; ++i) {
            double acc = p[i];
            if (factor)
                for (int j = 0; j < NX; ++j) acc += P[i * NX + j] * rb[k * NX + j];
            v[i] = acc;
        }
        for (int a = 0; a < NU; ++a) {
            double acc = Gk[NX + a];
            for (int i = 0; i < NX; ++i) acc += B[i * NU + a] * v[i];
            hu[a] = acc;
        }
        for (int c = 0; c < NX; ++c) {
            double acc = Gk[c];
            for (int i = 0; i < NX; ++i) acc += A[i * NX + c] * v[i];
            hx[c] = acc;
        }
        if (factor) {
            double PA[NX * NX], PB[NX * NU], Huu[4], Hux[NU * NX], Hxx[NX * NX];
            for (int i = 0; i < NX; ++i) {
                for (int c = 0; c < NX; ++c) {
                    double acc = 0.0;
                    for (int j = 0; j < NX; ++j) acc += P[i * NX + j] * A[j * NX + c];
                    PA[i * NX + c] = acc;
                }
                for (int a = 0; a < NU; ++a) {
                    double acc = 0.0;
                    for (int j = 0; j < NX; ++j) acc += P[i * NX + j] * B[j * NU + a];
                    PB[i * NU + a] = acc;
                }
            }
            for (int a = 0; a < NU; ++a) {
                for (int bb = 0; bb < NU; ++bb) {
                    double acc = H[(NX + a) * NZ + NX + bb];
                    for (int i = 0; i < NX; ++i) acc += B[i * NU + a] * PB[i * NU + bb];
                    Huu[a * NU + bb] = acc;
                }
                for (int c = 0; c < NX; ++c) {
                    double acc = H[(NX + a) * NZ + c];
                    for (int i = 0; i < NX; ++i) acc += B[i * NU + a] * PA[i * NX + c];
                    Hux[a * NX + c] = acc;
                }
            }
            for (int r = 0; r < NX; ++r)
                for (int c = 0; c < NX; ++c) {
                    double acc = H[r * NZ + c];
                    for (int i = 0; i < NX; ++i) acc += A[i * NX + r] * PA[i * NX + c];
                    Hxx[r * NX + c] = acc;
                }
            inv2(Huu, Hi);
            for (int a = 0; a < NU; ++a)
                for (int c = 0; c < NX; ++c)
                    K[a * NX + c] = -(Hi[a * NU + 0] * Hux[0 * NX + c] + Hi[a * NU + 1] * Hux[1 * NX + c]);
            for (int r = 0; r < NX; ++r)
                for (int c = 0; c < NX; ++c)
                    P[r * NX + c] = Hxx[r * NX + c] + Hux[0 * NX + r] * K[0 * NX + c] + Hux[1 * NX + r] * K[1 * NX + c];
            for (int r = 0; r < NX; ++r) /* symmetrise */
                for (int c = r + 1; c < NX; ++c) {
                    const double m = 0.5 * (P[r * NX + c] + P[c * NX + r]);
                    P[r * NX + c] = m;
                    P[c * NX + r] = m;
                }
        }
        kf[0] = -(Hi[0] * hu[0] + Hi[1] * hu[1]);
        kf[1] = -(Hi[2] * hu[0] + Hi[3] * hu[1]);
        for (int c = 0; c < NX; ++c) p[c] = hx[c] + K[0 * NX + c] * hu[0] + K[1 * NX + c] * hu[1];
    }
    /* forward */
    memset(dz, 0, sizeof(double) * (size_t)(N + 1) * NZ);
    for (int k = 0; k < N; ++k) {
        const double *A = qp->A + k * 36, *B = qp->B + k * 12;
        const double *K = w->K + k * 12, *kf = w->kff + k * 2;
        double *d = dz + k * NZ, *dn = dz + (k + 1) * NZ;
        for (int a = 0; a < NU; ++a) {
            double acc = kf[a];
            for (int c = 0; c < NX; ++c) acc += K[a * NX + c] * d[c];
            d[NX + a] = acc;
        }
        for (int i = 0; i < NX; ++i) {
            double acc = factor ? rb[k * NX + i] : 0.0;
            for (int c = 0; c < NX; ++c) acc += A[i * NX + c] * d[c];
            for (int a = 0; a < NU; ++a) acc += B[i * NU + a] * d[NX + a];
            dn[i] = acc;
        }
    }
}

/* Work areas: four per-thread blocks that grow to the largest request and are zeroed on every use
 * (what calloc gave), so a batch of solves does not allocate per solve. */
static __thread double *orc_ws_blk[4];
static __thread size_t orc_ws_cap[4];
static double *orc_ws_get(int slot, size_t n)
{
    if (orc_ws_cap[slot] < n) {
        free(orc_ws_blk[slot]);
        orc_ws_blk[slot] = (double *)malloc(n * sizeof(double));
        orc_ws_cap[slot] = orc_ws_blk[slot] ? n : 0;
    }
    if (orc_ws_blk[slot]) memset(orc_ws_blk[slot], 0, n * sizeof(double));
    return orc_ws_blk[slot];
}


/* ---------------------------------------------------------------- QP: active-set polish
 * The interior-point iteration identifies the active set long before it has converged; from there the
 * QP is finished exactly: the bounds classified active are imposed as equalities, the others dropped,
 * and the equality-constrained QP is solved by ONE Riccati recursion in which the pinned quantities are
 * eliminated stage by stage (no penalties).  The result is accepted if it satisfies the KKT conditions
 * of the original QP (multiplier signs of the pins, feasibility of the dropped bounds); otherwise the
 * working set is corrected by the primal-dual active-set rule and the solve repeated, and after
 * polish_tries corrections the interior-point iteration resumes where it was.
 *
 * ws[k*4+j]: -1 pair j of stage k at its lower bound, +1 at its upper bound, 0 free.
 * Stage k carries these equality rows C u + E x = d on (x_k, u_k):
 *   input pin a:             u_a = bound
 *   state pin (node k+1, s): B[s,:] u + A[s,:] x = bound - b_s
 * Returns 0, or -1 when the rows of some stage are not independent (an input pinned together with the
 * state it integrates into): the caller then leaves the working set alone.
 * z[(N+1)*8]: in: z_0[0..5] = dx0; out: minimiser on the working set.  sm[N*4]: signed multipliers of
 * the pins (stationarity: (Hz+g)_i + ... + sm_i = 0; an upper bound wants sm >= 0, a lower sm <= 0). */
int orc_ws_solve(const orc_qp *qp, const double *lo, const double *up, const int *ws, double *z, double *sm)
{
    const int N = qp->N;
    double *mem = orc_ws_get(2, (size_t)N * (12 + 2 + 4 * 2 + 4 * 6 + 4 + 1));
    if (!mem) return -1;
    double *Kall = mem, *kall = Kall + (size_t)N * 12, *Call = kall + (size_t)N * 2, *Eall = Call + (size_t)N * 8;
    double *dall = Eall + (size_t)N * 24, *mall = dall + (size_t)N * 4;
    double P[NX * NX], p[NX];
    memset(P, 0, sizeof(P));
    memset(p, 0, sizeof(p));
    for (int k = N - 1; k >= 0; --k) {
        const double *A = qp->A + k * 36, *B = qp->B + k * 12, *H = qp->H + k * 64, *g = qp->g + k * 8, *b = qp->b + k * 6;
        double *K = Kall + k * 12, *kf = kall + k * 2, *C = Call + k * 8, *E = Eall + k * 24, *d = dall + k * 4;
        /* equality rows of this stage */
        int m = 0;
        for (int a = 0; a < NU; ++a)
            if (ws[k * NB + a] != 0) {
                if (m == 2) return -1;
                C[m * 2] = a == 0; C[m * 2 + 1] = a == 1;
                for (int c = 0; c < NX; ++c) E[m * 6 + c] = 0.0;
                d[m] = ws[k * NB + a] < 0 ? lo[k * NB + a] : up[k * NB + a];
                ++m;
            }
        if (k + 1 < N)
            for (int j = 2; j < NB; ++j)
                if (ws[(k + 1) * NB + j] != 0) {
                    const int s = BIDX[j];   /* 4, 5 */
                    if (m == 2) return -1;
                    C[m * 2] = B[s * NU]; C[m * 2 + 1] = B[s * NU + 1];
                    for (int c = 0; c < NX; ++c) E[m * 6 + c] = A[s * NX + c];
                    d[m] = (ws[(k + 1) * NB + j] < 0 ? lo[(k + 1) * NB + j] : up[(k + 1) * NB + j]) - b[s];
                    ++m;
                }
        mall[k] = (double)m;
        /* stage quantities with the cost-to-go of the next node */
        double v[NX], PA[NX * NX], PB[NX * NU], Huu[4], Hux[NU * NX], Hxx[NX * NX], hu[NU], hx[NX];
        for (int i = 0; i < NX; ++i) {
            double acc = p[i];
            for (int j = 0; j < NX; ++j) acc += P[i * NX + j] * b[j];
            v[i] = acc;
            for (int c = 0; c < NX; ++c) {
                double s2 = 0.0;
                for (int j = 0; j < NX; ++j) s2 += P[i * NX + j] * A[j * NX + c];
                PA[i * NX + c] = s2;
            }
            for (int a = 0; a < NU; ++a) {
                double s2 = 0.0;
                for (int j = 0; j < NX; ++j) s2 += P[i * NX + j] * B[j * NU + a];
                PB[i * NU + a] = s2;
            }
        }
        for (int a = 0; a < NU; ++a) {
            double acc = g[NX + a];
            for (int i = 0; i < NX; ++i) acc += B[i * NU + a] * v[i];
            hu[a] = acc;
            for (int bb = 0; bb < NU; ++bb) {
                double s2 = H[(NX + a) * NZ + NX + bb];
                for (int i = 0; i < NX; ++i) s2 += B[i * NU + a] * PB[i * NU + bb];
                Huu[a * NU + bb] = s2;
            }
            for (int c = 0; c < NX; ++c) {
                double s2 = H[(NX + a) * NZ + c];
                for (int i = 0; i < NX; ++i) s2 += B[i * NU + a] * PA[i * NX + c];
                Hux[a * NX + c] = s2;
            }
        }
        for (int r = 0; r < NX; ++r) {
            double acc = g[r];
            for (int i = 0; i < NX; ++i) acc += A[i * NX + r] * v[i];
            hx[r] = acc;
            for (int c = 0; c < NX; ++c) {
                double s2 = H[r * NZ + c];
                for (int i = 0; i < NX; ++i) s2 += A[i * NX + r] * PA[i * NX + c];
                Hxx[r * NX + c] = s2;
            }
        }
        /* policy u = K x + kf of this stage */
        if (m == 0) {
            double Hi[4];
            inv2(Huu, Hi);
            for (int a = 0; a < NU; ++a) {
                for (int c = 0; c < NX; ++c) K[a * NX + c] = -(Hi[a * NU] * Hux[c] + Hi[a * NU + 1] * Hux[NX + c]);
                kf[a] = -(Hi[a * NU] * hu[0] + Hi[a * NU + 1] * hu[1]);
            }
        } else if (m == 1) {
            const double cc = C[0] * C[0] + C[1] * C[1];
            if (!(cc > 1e-300)) return -1;
            const double n0 = -C[1], n1 = C[0];                 /* null direction of the row */
            double Ux[NU * NX], u0[NU];
            for (int a = 0; a < NU; ++a) {
                for (int c = 0; c < NX; ++c) Ux[a * NX + c] = -C[a] * E[c] / cc;
                u0[a] = C[a] * d[0] / cc;
            }
            const double Hww = n0 * (Huu[0] * n0 + Huu[1] * n1) + n1 * (Huu[2] * n0 + Huu[3] * n1);
            double Kw[NX], kw;
            for (int c = 0; c < NX; ++c) {
                const double t0 = Huu[0] * Ux[c] + Huu[1] * Ux[NX + c] + Hux[c];
                const double t1 = Huu[2] * Ux[c] + Huu[3] * Ux[NX + c] + Hux[NX + c];
                Kw[c] = -(n0 * t0 + n1 * t1) / Hww;
            }
            kw = -(n0 * (Huu[0] * u0[0] + Huu[1] * u0[1] + hu[0]) + n1 * (Huu[2] * u0[0] + Huu[3] * u0[1] + hu[1])) / Hww;
            for (int c = 0; c < NX; ++c) { K[c] = Ux[c] + n0 * Kw[c]; K[NX + c] = Ux[NX + c] + n1 * Kw[c]; }
            kf[0] = u0[0] + n0 * kw;
            kf[1] = u0[1] + n1 * kw;
        } else {
            const double det = C[0] * C[3] - C[1] * C[2];
            const double sc = (fabs(C[0]) + fabs(C[1])) * (fabs(C[2]) + fabs(C[3]));
            if (!(fabs(det) > 1e-12 * sc)) return -1;
            const double i00 = C[3] / det, i01 = -C[1] / det, i10 = -C[2] / det, i11 = C[0] / det;
            for (int c = 0; c < NX; ++c) {
                K[c] = -(i00 * E[c] + i01 * E[6 + c]);
                K[NX + c] = -(i10 * E[c] + i11 * E[6 + c]);
            }
            kf[0] = i00 * d[0] + i01 * d[1];
            kf[1] = i10 * d[0] + i11 * d[1];
        }
        /* cost-to-go under that policy (general form, valid for any affine policy) */
        double HK[NU * NX], hk[NU], Pn[NX * NX], pn[NX];
        for (int a = 0; a < NU; ++a) {
            for (int c = 0; c < NX; ++c) HK[a * NX + c] = Huu[a * NU] * K[c] + Huu[a * NU + 1] * K[NX + c];
            hk[a] = hu[a] + Huu[a * NU] * kf[0] + Huu[a * NU + 1] * kf[1];
        }
        for (int r = 0; r < NX; ++r) {
            for (int c = 0; c < NX; ++c)
                Pn[r * NX + c] = Hxx[r * NX + c] + Hux[r] * K[c] + Hux[NX + r] * K[NX + c] + K[r] * Hux[c] + K[NX + r] * Hux[NX + c] +
                                 K[r] * HK[c] + K[NX + r] * HK[NX + c];
            pn[r] = hx[r] + Hux[r] * kf[0] + Hux[NX + r] * kf[1] + K[r] * hk[0] + K[NX + r] * hk[1];
        }
        for (int r = 0; r < NX; ++r) {
            p[r] = pn[r];
            for (int c = 0; c < NX; ++c) P[r * NX + c] = 0.5 * (Pn[r * NX + c] + Pn[c * NX + r]);
        }
    }
    /* forward */
    for (int k = 0; k < N; ++k) {
        const double *A = qp->A + k * 36, *B = qp->B + k * 12, *K = Kall + k * 12, *kf = kall + k * 2;
        double *zk = z + k * NZ, *zn = z + (k + 1) * NZ;
        for (int a = 0; a < NU; ++a) {
            double acc = kf[a];
            for (int c = 0; c < NX; ++c) acc += K[a * NX + c] * zk[c];
            zk[NX + a] = acc;
        }
        for (int i = 0; i < NX; ++i) {
            double acc = qp->b[k * NX + i];
            for (int c = 0; c < NX; ++c) acc += A[i * NX + c] * zk[c];
            for (int a = 0; a < NU; ++a) acc += B[i * NU + a] * zk[NX + a];
            zn[i] = acc;
        }
    }
    for (int a = 0; a < NU; ++a) z[N * NZ + NX + a] = 0.0;
    /* multipliers of the rows by the adjoint recursion: C' y = -(gradient wrt u), pi_k = gradient wrt x + E' y */
    double pi[NX];
    memset(pi, 0, sizeof(pi));
    memset(sm, 0, sizeof(double) * (size_t)N * NB);
    for (int k = N - 1; k >= 0; --k) {
        const double *A = qp->A + k * 36, *B = qp->B + k * 12, *H = qp->H + k * 64, *g = qp->g + k * 8, *zk = z + k * NZ;
        const double *C = Call + k * 8, *E = Eall + k * 24;
        const int m = (int)mall[k];
        double r[NZ], y[2] = {0.0, 0.0};
        for (int i = 0; i < NZ; ++i) {
            double acc = g[i];
            for (int j = 0; j < NZ; ++j) acc += H[i * NZ + j] * zk[j];
            r[i] = acc;
        }
        for (int c = 0; c < NX; ++c) for (int i = 0; i < NX; ++i) r[c] += A[i * NX + c] * pi[i];
        for (int a = 0; a < NU; ++a) for (int i = 0; i < NX; ++i) r[NX + a] += B[i * NU + a] * pi[i];
        if (m == 1) y[0] = -(C[0] * r[NX] + C[1] * r[NX + 1]) / (C[0] * C[0] + C[1] * C[1]);
        else if (m == 2) {
            const double det = C[0] * C[3] - C[1] * C[2];       /* C' y = -r_u */
            y[0] = -(C[3] * r[NX] - C[2] * r[NX + 1]) / det;
            y[1] = -(-C[1] * r[NX] + C[0] * r[NX + 1]) / det;
        }
        int row = 0;
        for (int a = 0; a < NU; ++a) if (ws[k * NB + a] != 0) sm[k * NB + a] = y[row++];
        if (k + 1 < N) for (int j = 2; j < NB; ++j) if (ws[(k + 1) * NB + j] != 0) sm[(k + 1) * NB + j] = y[row++];
        for (int c = 0; c < NX; ++c) {
            double acc = r[c];
            for (int q = 0; q < m; ++q) acc += E[q * 6 + c] * y[q];
            pi[c] = acc;
        }
    }
    return 0;
}

/* One polish attempt on the working set ws (in/out): solves, checks the KKT conditions of the QP and, while
 * they fail, corrects ws by the primal-dual active-set rule (at most `tries` times).  Returns 1 accepted
 * (z, sm hold the solution and its multipliers), 0 not accepted, -1 declined at once.  *solves counts
 * the Riccati recursions run. */
static int polish_attempt(const orc_config *cfg, const orc_qp *qp, const double *lo, const double *up, const double *msk,
                          int *ws, int tries, const double *z0, double *z, double *sm, int *solves)
{
    const int N = qp->N;
    const size_t nb = (size_t)N * NB;
    const double tol = cfg->polish_tol;
    /* one step of iterative refinement per solve: the same equality-constrained QP in increments from the
       first result (gradient, shooting gaps and bound distances re-evaluated there) */
    double *rm = orc_ws_get(3, (size_t)N * (8 + 6) + nb * 3 + (size_t)(N + 1) * NZ);
    if (!rm) return -1;
    double *g2 = rm, *b2 = g2 + (size_t)N * 8, *lo2 = b2 + (size_t)N * 6, *up2 = lo2 + nb, *rsm = up2 + nb, *dlt = rsm + nb;
    for (int tr = 0;; ++tr) {
        /* an input pinned together with the state it integrates into (node k+1) leaves stage k with dependent
           rows: keep the state's pin, free the input (if its bound then fails, the correction below says so) */
        for (int k = 0; k + 1 < N; ++k)
            for (int a = 0; a < NU; ++a)
                if (ws[k * NB + a] != 0 && ws[(k + 1) * NB + 2 + a] != 0) ws[k * NB + a] = 0;
        /* the solve in increments from the caller's point z0 (the interior-point iterate: a short step, so
           round-off stays relative to the step) */
        memcpy(z, z0, sizeof(double) * (size_t)(N + 1) * NZ);
        double gscale = 1.0;     /* size of the gradient entries the multipliers are differences of */
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) gscale = 1.0;
            for (int k = 0; k < N; ++k) {
                const double *A = qp->A + k * 36, *B = qp->B + k * 12, *H = qp->H + k * 64, *zk = z + k * NZ, *zn = z + (k + 1) * NZ;
                for (int i = 0; i < NZ; ++i) {
                    double acc = qp->g[k * NZ + i];
                    for (int j = 0; j < NZ; ++j) acc += H[i * NZ + j] * zk[j];
                    g2[k * NZ + i] = acc;
                    gscale = fmax(gscale, fabs(acc));
                }
                for (int i = 0; i < NX; ++i) {
                    double acc = qp->b[k * NX + i] - zn[i];
                    for (int c = 0; c < NX; ++c) acc += A[i * NX + c] * zk[c];
                    for (int a = 0; a < NU; ++a) acc += B[i * NU + a] * zk[NX + a];
                    b2[k * NX + i] = acc;
                }
                for (int j = 0; j < NB; ++j) {
                    const double v = zk[BIDX[j]];
                    lo2[k * NB + j] = lo[k * NB + j] - v;
                    up2[k * NB + j] = up[k * NB + j] - v;
                }
            }
            if (pass == 1) break;       /* (the second pass only evaluates the gradient at the result) */
            orc_qp q2 = *qp;
            q2.g = g2; q2.b = b2;
            memset(dlt, 0, sizeof(double) * (size_t)(N + 1) * NZ);     /* dx_0 increment: none */
            if (orc_ws_solve(&q2, lo2, up2, ws, dlt, sm) != 0) return tr == 0 ? -1 : 0;
            for (int i = 0; i < (N + 1) * NZ; ++i) z[i] += dlt[i];
        }
        ++*solves;
        const double told = tol * gscale;
        int bad = 0;
        for (size_t q = 0; q < nb; ++q) {
            if (msk[q] == 0.0) continue;
            const double v = z[(q / NB) * NZ + BIDX[q % NB]];
            if (ws[q] == 0) { if (!(v >= lo[q] - tol) || !(v <= up[q] + tol)) bad = 1; }
            else if (ws[q] < 0) { if (!(sm[q] <= told)) bad = 1; }
            else { if (!(sm[q] >= -told)) bad = 1; }
        }
        if (!bad) {
            /* accepted; one step of iterative refinement: the same solve once more, in increments from the
               result (g2, b2, lo2, up2 hold gradient, gaps and bound distances there) */
            orc_qp q2 = *qp;
            q2.g = g2; q2.b = b2;
            memset(dlt, 0, sizeof(double) * (size_t)(N + 1) * NZ);
            if (orc_ws_solve(&q2, lo2, up2, ws, dlt, rsm) == 0)
                for (int i = 0; i < (N + 1) * NZ; ++i) z[i] += dlt[i];
        }
        if (!bad) return 1;
        if (tr >= tries) return 0;
        for (size_t q = 0; q < nb; ++q) {
            if (msk[q] == 0.0) continue;
            const double v = z[(q / NB) * NZ + BIDX[q % NB]];
            if (ws[q] == 0) { if (!(v >= lo[q] - tol)) ws[q] = -1; else if (!(v <= up[q] + tol)) ws[q] = 1; }
            else if (ws[q] < 0) { if (!(sm[q] <= told)) ws[q] = 0; }
            else { if (!(sm[q] >= -told)) ws[q] = 0; }
        }
    }
}

int orc_qp_ipm(const orc_config *cfg, const orc_qp *qp, double *dx, double *du, double *lam,
               double *pi, int *iters, double *mu_out)
{
    return orc_qp_solve(cfg, qp, NULL, dx, du, lam, pi, iters, mu_out);
}

int orc_qp_solve(const orc_config *cfg, const orc_qp *qp, double *hint, double *dx, double *du, double *lam,
                 double *pi, int *iters, double *mu_out)
{
    const int N = qp->N;
    const size_t nb = (size_t)N * NB;
    double *mem = orc_ws_get(0, (size_t)(N + 1) * NZ * 4 + nb * 18 + (size_t)N * (64 + 8 + 6 + 8 + 12 + 4 + 2));
    if (!mem) return 4;
    double *z = mem, *dz = z + (N + 1) * NZ, *dzc = dz + (N + 1) * NZ, *zp = dzc + (N + 1) * NZ;
    double *tl = zp + (N + 1) * NZ, *tu = tl + nb, *ll = tu + nb, *lu = ll + nb;
    double *dtl = lu + nb, *dtu = dtl + nb, *dll = dtu + nb, *dlu = dll + nb;
    double *lo = dlu + nb, *up = lo + nb, *msk = up + nb, *rtl = msk + nb, *rtu = rtl + nb;
    double *sm = rtu + nb, *wsd = sm + nb, *ws0d = wsd + nb, *wspd = ws0d + nb, *clsd = wspd + nb;   /* polish: multipliers, working sets */
    double *Hd = clsd + nb, *G = Hd + (size_t)N * 64, *rb = G + (size_t)N * 8, *dG = rb + (size_t)N * 6;
    ric_ws w;
    w.N = N;
    w.K = dG + (size_t)N * 8;
    w.Hinv = w.K + (size_t)N * 12;
    w.kff = w.Hinv + (size_t)N * 4;

    int m = 0, status = 4, it = 0;
    for (int i = 0; i < NX; ++i) z[i] = qp->dx0[i];
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < NB; ++j) {
            const size_t q = (size_t)k * NB + j;
            if (j < 2) { lo[q] = qp->lu[k * 2 + j]; up[q] = qp->uu[k * 2 + j]; msk[q] = 1.0; }
            else { lo[q] = qp->lx[k * 2 + j - 2]; up[q] = qp->ux[k * 2 + j - 2]; msk[q] = (k >= 1) ? 1.0 : 0.0; }
            if (msk[q] != 0.0) {
                const double wv = z[k * NZ + BIDX[j]];
                tl[q] = fmax(wv - lo[q], cfg->ipm_thr0);
                tu[q] = fmax(up[q] - wv, cfg->ipm_thr0);
                ll[q] = cfg->ipm_mu0 / tl[q];
                lu[q] = cfg->ipm_mu0 / tu[q];
                m += 2;
            } else { tl[q] = tu[q] = 1.0; ll[q] = lu[q] = 0.0; lo[q] = -1.0; up[q] = 1.0; }
        }
    double lin_scale = 1.0, mu = 0.0;
    /* ---- active-set polish (see orc_ws_solve): working sets as int arrays carved from the work area */
    int *ws = (int *)wsd, *ws0 = (int *)ws0d, *wsprev = (int *)wspd, *cls = (int *)clsd;
    int n_pol = 0, polished = 0;
    for (size_t q = 0; q < nb; ++q) {
        /* the active set the iterate suggests: a bounded quantity sitting on its bound (the previous QP of a
           warm sequence left it there to round-off) */
        ws0[q] = 0;
        if (msk[q] != 0.0) { if (lo[q] >= -cfg->warm_eps) ws0[q] = -1; else if (up[q] <= cfg->warm_eps) ws0[q] = 1; }
        wsprev[q] = 2;
    }
    if (cfg->polish && hint && cfg->warm_tries >= 0 && *hint >= 1.0 && *hint <= 1.0 + cfg->warm_gate) {
        /* warm call whose predecessor ended on (nearly) the active set it started from: try that set first */
        memcpy(ws, ws0, nb * sizeof(int));
        if (polish_attempt(cfg, qp, lo, up, msk, ws, cfg->warm_tries, z, zp, sm, &n_pol) == 1) polished = 1;
    }
    for (it = 0; !polished; ++it) {
        mu = 0.0;
        for (size_t q = 0; q < nb; ++q) mu += msk[q] * (tl[q] * ll[q] + tu[q] * lu[q]);
        mu /= (double)m;
        if (!(mu == mu)) { status = 1; break; }
        /* pairwise complementarity: the average mu alone leaves an O(mu / lam_i) error on weakly
           active bounds (measured: 2e-4 at mu = 1e-10), so every pair must have collapsed */
        double cmax = 0.0;
        for (size_t q = 0; q < nb; ++q)
            cmax = fmax(cmax, fmax(fmin(tl[q], ll[q]), fmin(tu[q], lu[q])));
        if (it > 0 && mu <= cfg->ipm_tol_mu && lin_scale <= cfg->ipm_tol_lin && cmax <= cfg->ipm_tol_comp) { status = 0; break; }
        if (it >= cfg->ipm_max_iter) { status = (mu <= 1e-8 && lin_scale <= 1e-8) ? 0 : 4; break; }
        /* stalled: the step lengths have stayed tiny (infeasible or hopelessly started QP) */
        if (cfg->ipm_stall_iter > 0 && it >= cfg->ipm_stall_iter && lin_scale > cfg->ipm_stall_lin) { status = 4; break; }

        if (cfg->polish && it > 0 && mu <= cfg->polish_mu) {
            /* every pair classified by its slack against its multiplier, both as the last step predicted them
               at full length (see below); attempt when the classification is new */
            int same = 1;
            for (size_t q = 0; q < nb; ++q) {
                ws[q] = cls[q];
                if (ws[q] != wsprev[q]) same = 0;
            }
            if (!same) {
                memcpy(wsprev, ws, nb * sizeof(int));
                const int pr = polish_attempt(cfg, qp, lo, up, msk, ws, cfg->polish_tries, z, zp, sm, &n_pol);
                if (getenv("ORC_TRACE")) fprintf(stderr, "it %2d mu %.3e polish -> %d (solves so far %d)\n", it, mu, pr, n_pol);
                if (pr == 1) { polished = 1; break; }
            }
        }

        /* residuals and barrier-augmented Newton system */
        for (int k = 0; k < N; ++k) {
            const double *A = qp->A + k * 36, *B = qp->B + k * 12, *H = qp->H + k * 64;
            const double *zk = z + k * NZ, *zn = z + (k + 1) * NZ;
            for (int i = 0; i < NX; ++i) {
                double acc = qp->b[k * NX + i] - zn[i];
                for (int c = 0; c < NX; ++c) acc += A[i * NX + c] * zk[c];
                for (int a = 0; a < NU; ++a) acc += B[i * NU + a] * zk[NX + a];
                rb[k * NX + i] = acc;
            }
            memcpy(Hd + k * 64, H, sizeof(double) * 64);
            for (int i = 0; i < NZ; ++i) {
                double acc = qp->g[k * NZ + i];
                for (int j = 0; j < NZ; ++j) acc += H[i * NZ + j] * zk[j];
                G[k * NZ + i] = acc;
            }
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const int zi = BIDX[j];
                const double wv = zk[zi];
                rtl[q] = msk[q] * (wv - lo[q] - tl[q]);
                rtu[q] = msk[q] * (up[q] - wv - tu[q]);
                Hd[k * 64 + zi * NZ + zi] += ll[q] / tl[q] + lu[q] / tu[q];
                G[k * NZ + zi] += ll[q] / tl[q] * rtl[q] - lu[q] / tu[q] * rtu[q];
            }
        }
        /* predictor (affine scaling) direction */
        riccati_solve(qp, &w, Hd, G, rb, 1, dz);
        double a_aff = 1.0;
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const double dw = dz[k * NZ + BIDX[j]];
                dtl[q] = msk[q] * (dw + rtl[q]);
                dtu[q] = msk[q] * (-dw + rtu[q]);
                dll[q] = -ll[q] - ll[q] / tl[q] * dtl[q];
                dlu[q] = -lu[q] - lu[q] / tu[q] * dtu[q];
                if (dtl[q] < 0.0) a_aff = fmin(a_aff, -tl[q] / dtl[q]);
                if (dtu[q] < 0.0) a_aff = fmin(a_aff, -tu[q] / dtu[q]);
                if (dll[q] < 0.0) a_aff = fmin(a_aff, -ll[q] / dll[q]);
                if (dlu[q] < 0.0) a_aff = fmin(a_aff, -lu[q] / dlu[q]);
            }
        double mu_aff = 0.0;
        for (size_t q = 0; q < nb; ++q)
            mu_aff += msk[q] * ((tl[q] + a_aff * dtl[q]) * (ll[q] + a_aff * dll[q]) +
                                (tu[q] + a_aff * dtu[q]) * (lu[q] + a_aff * dlu[q]));
        mu_aff /= (double)m;
        const double sr = mu_aff / mu;
        const double smu = sr * sr * sr * mu; /* sigma * mu */
        /* second-order (Mehrotra) term damped when the affine step is almost fully blocked:
           an undamped correction then over-corrects and the iteration can cycle */
        const double csc = fmin(1.0, 4.0 * a_aff);
        /* corrector: only the bounded components of the gradient change */
        memset(dG, 0, sizeof(double) * (size_t)N * 8);
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const double cl = smu - csc * dtl[q] * dll[q], cu = smu - csc * dtu[q] * dlu[q];
                dG[k * NZ + BIDX[j]] = msk[q] * (-cl / tl[q] + cu / tu[q]);
            }
        riccati_solve(qp, &w, Hd, dG, rb, 0, dzc);
        double a_max = 1e300;
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < NZ; ++i) dz[k * NZ + i] += dzc[k * NZ + i];
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const double dw = dz[k * NZ + BIDX[j]];
                const double cl = smu - csc * dtl[q] * dll[q], cu = smu - csc * dtu[q] * dlu[q];
                const double ntl = msk[q] * (dw + rtl[q]), ntu = msk[q] * (-dw + rtu[q]);
                dll[q] = msk[q] * (cl / tl[q] - ll[q] - ll[q] / tl[q] * ntl);
                dlu[q] = msk[q] * (cu / tu[q] - lu[q] - lu[q] / tu[q] * ntu);
                dtl[q] = ntl;
                dtu[q] = ntu;
                if (dtl[q] < 0.0) a_max = fmin(a_max, -tl[q] / dtl[q]);
                if (dtu[q] < 0.0) a_max = fmin(a_max, -tu[q] / dtu[q]);
                if (dll[q] < 0.0) a_max = fmin(a_max, -ll[q] / dll[q]);
                if (dlu[q] < 0.0) a_max = fmin(a_max, -lu[q] / dlu[q]);
            }
        const double tau = fmax(cfg->ipm_tau, 1.0 - mu_aff);
        const double alpha = fmin(1.0, tau * a_max);
        /* active-set classification for the polish: where the full step (not cut back to stay interior) would
           leave slack below multiplier.  One step ahead of comparing the iterate's own pairs, and the limit the
           pair is heading for rather than where the step length stopped it. */
        for (size_t q = 0; q < nb; ++q) {
            cls[q] = 0;
            if (msk[q] != 0.0) {
                if (tl[q] + dtl[q] < ll[q] + dll[q]) cls[q] = -1;
                else if (tu[q] + dtu[q] < lu[q] + dlu[q]) cls[q] = 1;
            }
        }
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < NZ; ++i) z[k * NZ + i] += alpha * dz[k * NZ + i];
        for (size_t q = 0; q < nb; ++q) {
            tl[q] += alpha * dtl[q]; tu[q] += alpha * dtu[q];
            ll[q] += alpha * dll[q]; lu[q] += alpha * dlu[q];
        }
        lin_scale *= (1.0 - alpha);
        if (getenv("ORC_TRACE"))
            fprintf(stderr, "it %2d mu %.3e a_aff %.4f sigma %.3e alpha %.6f lin %.3e\n", it, mu, a_aff,
                    sr * sr * sr, alpha, lin_scale);
    }
    if (polished) {
        /* the vertex: bounds classified active hold to round-off, complementarity is exact */
        status = 0;
        memcpy(z, zp, sizeof(double) * (size_t)(N + 1) * NZ);
        int mism = 0;
        for (size_t q = 0; q < nb; ++q) {
            ll[q] = ws[q] < 0 ? fmax(-sm[q], 0.0) : 0.0;
            lu[q] = ws[q] > 0 ? fmax(sm[q], 0.0) : 0.0;
            mism += ws[q] != ws0[q];
        }
        if (hint) *hint = 1.0 + (double)mism;
        mu = 0.0;
    } else if (hint) *hint = 0.0;
    it += n_pol;   /* reported count: factorisations (interior-point iterations + active-set solves) */
    for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) dx[k * NX + i] = z[k * NZ + i];
    for (int k = 0; k < N; ++k) {
        du[k * NU + 0] = z[k * NZ + 6];
        du[k * NU + 1] = z[k * NZ + 7];
        if (lam)
            for (int j = 0; j < NB; ++j) {
                lam[k * 8 + j] = ll[k * NB + j];
                lam[k * 8 + 4 + j] = lu[k * NB + j];
            }
    }
    if (pi) { /* costates by the adjoint recursion of x-stationarity */
        memset(pi, 0, sizeof(double) * (size_t)(N + 1) * NX);
        for (int k = N - 1; k >= 1; --k) {
            const double *A = qp->A + k * 36, *H = qp->H + k * 64, *zk = z + k * NZ;
            for (int c = 0; c < NX; ++c) {
                double acc = qp->g[k * NZ + c];
                for (int j = 0; j < NZ; ++j) acc += H[c * NZ + j] * zk[j];
                for (int i = 0; i < NX; ++i) acc += A[i * NX + c] * pi[(k + 1) * NX + i];
                pi[k * NX + c] = acc;
            }
            pi[k * NX + 4] += -ll[k * NB + 2] + lu[k * NB + 2];
            pi[k * NX + 5] += -ll[k * NB + 3] + lu[k * NB + 3];
        }
    }
    if (iters) *iters = it;
    if (mu_out) *mu_out = mu;
    /* (the work area stays with the thread) */
    return status;
}

/* ---------------------------------------------------------------- SQP-RTI */

void orc_init_guess(const orc_config *c, const double *x0, double *xtraj, double *utraj)
{
    for (int k = 0; k <= c->N; ++k) memcpy(xtraj + k * NX, x0, sizeof(double) * NX);
    memset(utraj, 0, sizeof(double) * (size_t)c->N * NU);
}

typedef struct lin_ws {
    double *A, *B, *b, *H, *g, *lu, *uu, *lx, *ux, *dx, *du, *lam, *pi;
} lin_ws;

static int lin_alloc(lin_ws *w, int N)
{
    const size_t n = (size_t)N;
    double *m = orc_ws_get(1, n * (36 + 12 + 6 + 64 + 8 + 8 + 2 + 8) + (n + 1) * 12);
    if (!m) return 1;
    w->A = m; w->B = w->A + n * 36; w->b = w->B + n * 12; w->H = w->b + n * 6; w->g = w->H + n * 64;
    w->lu = w->g + n * 8; w->uu = w->lu + n * 2; w->lx = w->uu + n * 2; w->ux = w->lx + n * 2;
    w->du = w->ux + n * 2; w->lam = w->du + n * 2; w->dx = w->lam + n * 8; w->pi = w->dx + (n + 1) * 6;
    return 0;
}

static double linearise(const orc_config *c, const double *p, const double *xtraj,
                        const double *utraj, lin_ws *w)
{
    double cost = 0.0;
    for (int k = 0; k < c->N; ++k) {
        const double *x = xtraj + k * NX, *u = utraj + k * NU;
        double phi[NX];
        orc_rk4_sens(c, x, u, p, phi, w->A + k * 36, w->B + k * 12);
        for (int i = 0; i < NX; ++i) w->b[k * NX + i] = phi[i] - xtraj[(k + 1) * NX + i];
        orc_stage_gn(c, x, u, p, w->H + k * 64, w->g + k * 8);
        cost += stage_cost(c, x, u, p);
        for (int a = 0; a < 2; ++a) {
            w->lu[k * 2 + a] = c->lbu[a] - u[a];
            w->uu[k * 2 + a] = c->ubu[a] - u[a];
            w->lx[k * 2 + a] = c->lbx[a] - x[4 + a];
            w->ux[k * 2 + a] = c->ubx[a] - x[4 + a];
        }
    }
    return cost;
}

static int has_nan(const double *v, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!(v[i] == v[i]) || fabs(v[i]) > 1e300) return 1;
    return 0;
}

int orc_rti(const orc_config *c, const double *x0, const double *p, double *xtraj, double *utraj,
            orc_stats *st)
{
    return orc_rti_h(c, x0, p, xtraj, utraj, NULL, st);
}

int orc_rti_h(const orc_config *c, const double *x0, const double *p, double *xtraj, double *utraj,
              double *hint, orc_stats *st)
{
    const int N = c->N;
    lin_ws w;
    orc_stats local;
    if (!st) st = &local;
    memset(st, 0, sizeof(*st));
    if (lin_alloc(&w, N)) { st->status = 4; return 4; }
    st->cost = linearise(c, p, xtraj, utraj, &w);
    double dx0[NX];
    for (int i = 0; i < NX; ++i) dx0[i] = x0[i] - xtraj[i];
    orc_qp qp = {N, w.A, w.B, w.b, w.H, w.g, w.lu, w.uu, w.lx, w.ux, dx0};
    int iters = 0;
    int qs = orc_qp_solve(c, &qp, hint, w.dx, w.du, w.lam, w.pi, &iters, NULL);
    st->ipm_iters = iters;
    st->sqp_iters = 1;
    if (qs == 0 && (has_nan(w.dx, (size_t)(N + 1) * NX) || has_nan(w.du, (size_t)N * NU))) qs = 1;
    if (qs == 1) {
        st->status = 1; /* iterate left untouched */
    } else {
        double s = 0.0;
        for (int i = 0; i < (N + 1) * NX; ++i) { xtraj[i] += w.dx[i]; s = fmax(s, fabs(w.dx[i])); }
        for (int i = 0; i < N * NU; ++i) { utraj[i] += w.du[i]; s = fmax(s, fabs(w.du[i])); }
        st->step_inf = s;
        st->status = qs;
    }
    return st->status;
}

int orc_sqp(const orc_config *c, const double *x0, const double *p, double *xtraj, double *utraj,
            int max_iter, double tol, orc_stats *st)
{
    return orc_sqp_h(c, x0, p, xtraj, utraj, max_iter, tol, NULL, st);
}

int orc_sqp_h(const orc_config *c, const double *x0, const double *p, double *xtraj, double *utraj,
              int max_iter, double tol, double *hint, orc_stats *st)
{
    const int N = c->N;
    lin_ws w;
    orc_stats local;
    if (!st) st = &local;
    memset(st, 0, sizeof(*st));
    if (lin_alloc(&w, N)) { st->status = 4; return 4; }
    int status = 2;
    double own_hint = 0.0;   /* successive QPs of one SQP call are a warm sequence of their own */
    if (!hint) hint = &own_hint;
    for (int it = 0;; ++it) {
        st->cost = linearise(c, p, xtraj, utraj, &w);
        /* NLP KKT residuals at the current iterate with the last QP's multipliers */
        double rs = 0.0, re = 0.0, ri = 0.0, rc = 0.0;
        for (int i = 0; i < NX; ++i) re = fmax(re, fabs(x0[i] - xtraj[i]));
        for (int k = 0; k < N; ++k) {
            const double *A = w.A + k * 36, *B = w.B + k * 12;
            for (int i = 0; i < NX; ++i) re = fmax(re, fabs(w.b[k * NX + i]));
            for (int cc = 0; cc < NX && k >= 1; ++cc) {
                double acc = w.g[k * 8 + cc] - w.pi[k * NX + cc];
                for (int i = 0; i < NX; ++i) acc += A[i * NX + cc] * w.pi[(k + 1) * NX + i];
                if (cc >= 4) acc += -w.lam[k * 8 + cc - 2] + w.lam[k * 8 + 4 + cc - 2];
                rs = fmax(rs, fabs(acc));
            }
            for (int a = 0; a < NU; ++a) {
                double acc = w.g[k * 8 + NX + a] - w.lam[k * 8 + a] + w.lam[k * 8 + 4 + a];
                for (int i = 0; i < NX; ++i) acc += B[i * NU + a] * w.pi[(k + 1) * NX + i];
                rs = fmax(rs, fabs(acc));
            }
            for (int j = 0; j < NB; ++j) {
                if (j >= 2 && k == 0) continue;
                const double sl = (j < 2) ? -w.lu[k * 2 + j] : -w.lx[k * 2 + j - 2]; /* value - lb */
                const double su = (j < 2) ? w.uu[k * 2 + j] : w.ux[k * 2 + j - 2];   /* ub - value */
                ri = fmax(ri, fmax(-sl, -su));
                rc = fmax(rc, fabs(w.lam[k * 8 + j] * sl));
                rc = fmax(rc, fabs(w.lam[k * 8 + 4 + j] * su));
            }
        }
        st->res_stat = rs; st->res_eq = re; st->res_ineq = ri; st->res_comp = rc;
        if (it > 0 && rs <= tol && re <= tol && ri <= tol && rc <= tol) { status = 0; break; }
        if (it >= max_iter) { status = 2; break; }
        double dx0[NX];
        for (int i = 0; i < NX; ++i) dx0[i] = x0[i] - xtraj[i];
        orc_qp qp = {N, w.A, w.B, w.b, w.H, w.g, w.lu, w.uu, w.lx, w.ux, dx0};
        int iters = 0;
        int qs = orc_qp_solve(c, &qp, hint, w.dx, w.du, w.lam, w.pi, &iters, NULL);
        st->ipm_iters += iters;
        st->sqp_iters = it + 1;
        if (qs == 0 && (has_nan(w.dx, (size_t)(N + 1) * NX) || has_nan(w.du, (size_t)N * NU))) qs = 1;
        if (qs != 0) { status = qs; if (qs == 1) break; }
        double s = 0.0;
        for (int i = 0; i < (N + 1) * NX; ++i) { xtraj[i] += w.dx[i]; s = fmax(s, fabs(w.dx[i])); }
        for (int i = 0; i < N * NU; ++i) { utraj[i] += w.du[i]; s = fmax(s, fabs(w.du[i])); }
        st->step_inf = s;
        if (qs != 0) break;
    }
    st->status = status;
    return status;
}

void orc_rti_batch(const orc_config *c, int Bn, const double *x0, const double *p, double *xtraj,
                   double *utraj, int *status, int *ipm_iters, int cold)
{
    orc_rti_batch_h(c, Bn, x0, p, xtraj, utraj, NULL, status, ipm_iters, cold);
}

void orc_rti_batch_h(const orc_config *c, int Bn, const double *x0, const double *p, double *xtraj,
                     double *utraj, double *hint, int *status, int *ipm_iters, int cold)
{
    const size_t sx = (size_t)(c->N + 1) * NX, su = (size_t)c->N * NU;
    for (int i = 0; i < Bn; ++i) {
        orc_stats st;
        if (cold) orc_init_guess(c, x0 + (size_t)i * NX, xtraj + i * sx, utraj + i * su);
        if (cold && hint) hint[i] = 0.0;
        orc_rti_h(c, x0 + (size_t)i * NX, p + (size_t)i * ORC_NP, xtraj + i * sx, utraj + i * su, hint ? hint + i : NULL, &st);
        if (hint && st.status != 0) hint[i] = 0.0;
        if (status) status[i] = st.status;
        if (ipm_iters) ipm_iters[i] = st.ipm_iters;
    }
}

/* The same batch with one instance per OpenMP thread (bench.py's cpu_baseline leg, built with
 * -fopenmp; without OpenMP it is the serial loop). */
void orc_rti_batch_mt(const orc_config *c, int Bn, const double *x0, const double *p, double *xtraj,
                      double *utraj, int *status, int *ipm_iters, int cold, int nthreads)
{
    orc_rti_batch_mt_h(c, Bn, x0, p, xtraj, utraj, NULL, status, ipm_iters, cold, nthreads);
}

void orc_rti_batch_mt_h(const orc_config *c, int Bn, const double *x0, const double *p, double *xtraj,
                        double *utraj, double *hint, int *status, int *ipm_iters, int cold, int nthreads)
{
    const size_t sx = (size_t)(c->N + 1) * NX, su = (size_t)c->N * NU;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
    for (int i = 0; i < Bn; ++i) {
        orc_stats st;
        if (cold) orc_init_guess(c, x0 + (size_t)i * NX, xtraj + i * sx, utraj + i * su);
        if (cold && hint) hint[i] = 0.0;
        orc_rti_h(c, x0 + (size_t)i * NX, p + (size_t)i * ORC_NP, xtraj + i * sx, utraj + i * su, hint ? hint + i : NULL, &st);
        if (hint && st.status != 0) hint[i] = 0.0;
        if (status) status[i] = st.status;
        if (ipm_iters) ipm_iters[i] = st.ipm_iters;
    }
}

/* ---------------------------------------------------------------- path front-end */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif

int orc_test_track(double *xy, int cap)
{
    /* utilities.cpp:59-72 : the fp accumulation of theta decides the point count */
    const double n = 400.0;
    int cnt = 0;
    for (double theta = 0; theta < 2 * M_PI; theta += 2 * M_PI / n) {
        const double radius = 20;
        if (cnt < cap) {
            xy[2 * cnt] = 2 * radius * cos(theta);
            xy[2 * cnt + 1] = radius * sin(1 * theta) + radius / 1.2 * sin(3 * theta);
        }
        ++cnt;
    }
    return cnt;
}

static double dist_sqrd(double dx, double dy) { return dx * dx + dy * dy; }

void orc_track_section(const double *xy, int n, double sx, double sy, int *start, int *end)
{
    /* PathTrackingSys.cpp:107-136 */
    const double maxLen = 15;
    double best = dist_sqrd(sx - xy[0], sy - xy[1]);
    int mi = 0;
    for (int i = 1; i < n; ++i) {
        const double d = dist_sqrd(sx - xy[2 * i], sy - xy[2 * i + 1]);
        if (d < best) { best = d; mi = i; }
    }
    double len = 0;
    int fi = mi;
    while (len < maxLen && fi < n - 1) {
        ++fi;
        len += sqrt(dist_sqrd(xy[2 * fi] - xy[2 * (fi - 1)], xy[2 * fi + 1] - xy[2 * (fi - 1) + 1]));
    }
    int s = mi, e = fi;
    if (e - s < 4) e = s + 4;
    if (e >= n) { s = 0; e = s + 4; }
    *start = s;
    *end = e;
}

void orc_polyfit3(const double *xs, const double *ys, int n, double *coeffs)
{
    /* least squares on the n x 4 Vandermonde matrix by unpivoted Householder QR
       (utilities.cpp:39-57 uses Eigen householderQr().solve()) */
    double *A = (double *)malloc(sizeof(double) * (size_t)n * 5);
    for (int i = 0; i < n; ++i) {
        A[i * 5 + 0] = 1.0;
        for (int j = 0; j < 3; ++j) A[i * 5 + j + 1] = A[i * 5 + j] * xs[i];
        A[i * 5 + 4] = ys[i];
    }
    for (int j = 0; j < 4 && j < n; ++j) {
        double nrm = 0.0;
        for (int i = j; i < n; ++i) nrm += A[i * 5 + j] * A[i * 5 + j];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = (A[j * 5 + j] > 0) ? -nrm : nrm;
        const double v0 = A[j * 5 + j] - alpha;
        double vnorm2 = v0 * v0;
        for (int i = j + 1; i < n; ++i) vnorm2 += A[i * 5 + j] * A[i * 5 + j];
        if (vnorm2 == 0.0) continue;
        for (int cidx = j + 1; cidx < 5; ++cidx) {
            double dot = v0 * A[j * 5 + cidx];
            for (int i = j + 1; i < n; ++i) dot += A[i * 5 + j] * A[i * 5 + cidx];
            const double f = 2.0 * dot / vnorm2;
            A[j * 5 + cidx] -= f * v0;
            for (int i = j + 1; i < n; ++i) A[i * 5 + cidx] -= f * A[i * 5 + j];
        }
        A[j * 5 + j] = alpha;
        for (int i = j + 1; i < n; ++i) A[i * 5 + j] = 0.0; /* v no longer needed */
    }
    for (int j = 3; j >= 0; --j) {
        double acc = A[j * 5 + 4];
        for (int cidx = j + 1; cidx < 4; ++cidx) acc -= A[j * 5 + cidx] * coeffs[cidx];
        coeffs[j] = acc / A[j * 5 + j];
    }
    free(A);
}

static double fit_rotation(const double *xy, const double *state, double rot, int start, int end,
                           double *coeffs)
{
    /* PathTrackingSys.cpp:81-105 */
    const int n = end - start;
    double *xs = (double *)malloc(sizeof(double) * (size_t)n * 2), *ys = xs + n;
    const double angle = rot - state[2];
    for (int i = start; i < end; ++i) {
        const double dx = xy[2 * i] - state[0], dy = xy[2 * i + 1] - state[1];
        xs[i - start] = dx * cos(angle) - dy * sin(angle);
        ys[i - start] = dx * sin(angle) + dy * cos(angle);
    }
    orc_polyfit3(xs, ys, n, coeffs);
    double cost = 0;
    for (int i = 0; i < n; ++i) {
        const double x = xs[i];
        const double e = ys[i] - (coeffs[0] + coeffs[1] * x + coeffs[2] * x * x + coeffs[3] * x * x * x);
        cost += e * e + 0.0;
    }
    free(xs);
    return cost;
}

double orc_calc_coeffs(const double *xy, int n, const double *state, double *coeffs)
{
    /* PathTrackingSys.cpp:63-79 : fp loop gives 4 rotations, strict '<' keeps the first minimum */
    int start, end;
    orc_track_section(xy, n, state[0], state[1], &start, &end);
    double minCost = 1e19, rotation = 0;
    for (double rot = -M_PI_2; rot < 0; rot += M_PI_2 / 3) {
        double cur[4];
        const double cost = fit_rotation(xy, state, rot, start, end, cur);
        if (cost < minCost) {
            minCost = cost;
            memcpy(coeffs, cur, sizeof(cur));
            rotation = rot;
        }
    }
    return rotation;
}
