/*
 * nmpc_oracle.h -- CPU restatement (plain C99, fp64, serial) of the reference's
 * path-tracking NMPC hot path.  TEST INFRASTRUCTURE ONLY: it may be used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
 * product path under mpc_local_planner_amd/.
 *
 * Parity status: the reference's arithmetic lives in acados / HPIPM / BLASFEO /
 * CasADi (un-vendored, unpinned: /root/reference/install_acados.sh:38-54) and in
 * generated C that is git-ignored (/root/reference/.gitignore:4-5), and the
 * reference holds no golden vectors => **acados parity unpinned**.  This oracle
 * restates the documented algorithm from the reference's own problem statement:
 *   model      /root/reference/acados/mpcPathTracking.py:41-47
 *   parameters /root/reference/acados/mpcPathTracking.py:51
 *   cost       /root/reference/acados/mpcPathTracking.py:63-80,100-103
 *   bounds     /root/reference/acados/mpcPathTracking.py:105-115
 *   options    /root/reference/acados/mpcPathTracking.py:123-128
 *   values     /root/reference/config/mpc_params.yaml:35-59
 *   protocol   /root/reference/src/AcadosSolver.cpp:15-92
 * and is itself pinned by mpmath/sympy fixtures (tests/golden/) and by KKT
 * certificates computed independently in tests/.
 */
#ifndef NMPC_ORACLE_H
#define NMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 6
#define ORC_NU 2
#define ORC_NP 6
#define ORC_NY 7
#define ORC_NZ 8 /* stage variable ordering in this oracle: [x(6); u(2)] */

typedef struct orc_config {
    int N;             /* horizon (mpc_params.yaml:35) */
    double dt;         /* mpc_params.yaml:41 */
    double wheelbase;  /* mpc_params.yaml:59 */
    double W[ORC_NY];  /* mpcPathTracking.py:103 */
    double lbu[2], ubu[2]; /* steering_angle_dot_range, throttle_dot_range */
    double lbx[2], ubx[2]; /* steering_angle_range, throttle_range (states 4,5) */
    /* interior-point parameters (shared with the HIP path so both walk the same
       iteration sequence) */
    int ipm_max_iter;
    double ipm_mu0;
    double ipm_thr0;
    double ipm_tau;
    double ipm_tol_mu;
    double ipm_tol_lin;
    double ipm_tol_comp; /* max_i min(t_i, lam_i): pairwise complementarity, see orc_qp_ipm */
    int ipm_stall_iter;  /* early failure (status 4): after this many iterations ... (0: never) */
    double ipm_stall_lin; /* ... the infeasibility scale is still above this */
    /* active-set polish of the QP (see orc_ws_solve) */
    int polish;           /* 0: interior point to the end; 1: finish on the identified active set */
    double polish_mu;     /* first attempt once the average complementarity is below this */
    int polish_tries;     /* primal-dual active-set corrections per attempt */
    double polish_tol;    /* KKT tolerance of acceptance (multiplier signs, feasibility of dropped bounds) */
    int warm_tries;       /* warm calls: corrections of the attempt on the iterate's own active set; < 0: never try it */
    int warm_gate;        /* ... tried when the previous call moved at most this many pairs in or out of the set */
    double warm_eps;      /* iterate within this of a bound = guessed active */
} orc_config;

/* fills cfg with the reference's defaults (N and dt given explicitly) */
void orc_default_config(orc_config *cfg, int N, double dt);

/* continuous model f(x,u,p) (mpcPathTracking.py:41-47) */
void orc_f(const orc_config *cfg, const double *x, const double *u, const double *p, double *f);

/* one classical RK4 step of size dt with forward sensitivities (acados ERK,
   4 stages, 1 step): phi[6], A[6*6] row-major = dphi/dx, B[6*2] = dphi/du */
void orc_rk4_sens(const orc_config *cfg, const double *x, const double *u, const double *p,
                  double *phi, double *A, double *B);

/* NLS output y[7] and Jacobian J[7*8] row-major, columns ordered [x(6), u(2)]
   (mpcPathTracking.py:63-80) */
void orc_cost_yJ(const orc_config *cfg, const double *x, const double *u, const double *p,
                 double *y, double *J);

/* Gauss-Newton stage terms: H[8*8] = dt J'WJ, g[8] = dt J'W y, ordered [x;u] */
void orc_stage_gn(const orc_config *cfg, const double *x, const double *u, const double *p,
                  double *H, double *g);

/* OCP-structured QP in increments (dx_k, du_k):
 *   min  sum_{k<N} 1/2 z_k' H_k z_k + g_k' z_k ,  z_k = [dx_k; du_k]
 *   s.t. dx_{k+1} = A_k dx_k + B_k du_k + b_k,  dx_0 given
 *        lu_k <= du_k <= uu_k            (k = 0..N-1)
 *        lx_k <= dx_k[4,5] <= ux_k       (k = 1..N-1)
 * solved by a Mehrotra predictor-corrector interior-point method whose Newton
 * systems are solved by a Riccati recursion.  Outputs dx[(N+1)*6], du[N*2],
 * bound multipliers lam[N*8] (per stage: lower u0,u1,x4,x5 then upper ...),
 * costates pi[(N+1)*6] (pi_0 unused).  Returns 0 on convergence, 4 on
 * QP failure (acados numbering), 1 on NaN.
 */
typedef struct orc_qp {
    int N;
    const double *A;  /* N*36 */
    const double *B;  /* N*12 */
    const double *b;  /* N*6  */
    const double *H;  /* N*64, [x;u] ordering */
    const double *g;  /* N*8 */
    const double *lu, *uu; /* N*2 */
    const double *lx, *ux; /* N*2 (entry k=0 ignored) */
    const double *dx0;     /* 6 */
} orc_qp;

int orc_qp_ipm(const orc_config *cfg, const orc_qp *qp, double *dx, double *du, double *lam,
               double *pi, int *iters, double *mu_out);
/* The equality-constrained QP on a working set, solved exactly by one Riccati recursion in which the pinned
 * quantities are eliminated stage by stage (what the active-set polish of orc_qp_solve runs).
 * ws[N*4]: per stage the pairs u0, u1, x4, x5: -1 at the lower bound lo[], +1 at the upper bound up[], 0 free
 * (state pairs of stage 0 are ignored).  z[(N+1)*8] per node [dx(6); du(2)]: in z[0..5] = dx0, out the
 * minimiser; sm[N*4]: signed multipliers of the pins (an upper bound wants >= 0, a lower <= 0).
 * Returns 0, or -1 when a stage's pins are not independent (an input pinned together with the state of the
 * next node it integrates into). */
int orc_ws_solve(const orc_qp *qp, const double *lo, const double *up, const int *ws, double *z, double *sm);

/* the same with the warm-start hint of a sequence of QPs (in/out, one double per instance that the caller
   keeps between calls and zeroes on a cold start; NULL: none): 0 = nothing known, 1 + n = the previous QP
   ended polished on an active set that differed from the one its iterate suggested in n pairs */
int orc_qp_solve(const orc_config *cfg, const orc_qp *qp, double *hint, double *dx, double *du, double *lam,
                 double *pi, int *iters, double *mu_out);

typedef struct orc_stats {
    int status;       /* 0 ok, 1 NaN, 2 max iter (SQP), 4 QP failure */
    int sqp_iters;
    int ipm_iters;    /* total over SQP iterations */
    double cost;      /* sum_k dt/2 ||y_k||_W^2 at the returned iterate's linearisation point */
    double res_stat, res_eq, res_ineq, res_comp; /* NLP residuals (SQP mode) */
    double step_inf;  /* ||dz||_inf of last step */
} orc_stats;

/* one SQP-RTI call (AcadosSolver.cpp:28-76 + acados ocp_nlp_sqp_rti, phase 0):
   xtraj[(N+1)*6], utraj[N*2] in/out (warm start, no shift: PathTrackingSys.cpp:58-61) */
int orc_rti(const orc_config *cfg, const double *x0, const double *p, double *xtraj,
            double *utraj, orc_stats *st);
/* ... with the warm-start hint of orc_qp_solve carried from call to call (one double, zero it on a cold
   start and after a failed call) */
int orc_rti_h(const orc_config *cfg, const double *x0, const double *p, double *xtraj,
              double *utraj, double *hint, orc_stats *st);

/* full SQP: repeat until the NLP KKT residuals <= tol or max_iter */
int orc_sqp(const orc_config *cfg, const double *x0, const double *p, double *xtraj, double *utraj,
            int max_iter, double tol, orc_stats *st);
int orc_sqp_h(const orc_config *cfg, const double *x0, const double *p, double *xtraj, double *utraj,
              int max_iter, double tol, double *hint, orc_stats *st);

/* cold guess x_k = x0, u_k = 0 (AcadosSolver.cpp:78-92) */
void orc_init_guess(const orc_config *cfg, const double *x0, double *xtraj, double *utraj);

/* batch helpers (serial loop; used for the cpu_baseline leg and parity tests).
   x0[B*6], p[B*6], xtraj[B*(N+1)*6], utraj[B*N*2], status[B], ipm_iters[B] */
void orc_rti_batch(const orc_config *cfg, int B, const double *x0, const double *p, double *xtraj,
                   double *utraj, int *status, int *ipm_iters, int cold);
/* one instance per OpenMP thread (serial without -fopenmp) */
void orc_rti_batch_mt(const orc_config *cfg, int B, const double *x0, const double *p, double *xtraj,
                      double *utraj, int *status, int *ipm_iters, int cold, int nthreads);
/* the same two with the per-instance warm-start hints hint[B] (cold != 0 zeroes them) */
void orc_rti_batch_h(const orc_config *cfg, int B, const double *x0, const double *p, double *xtraj,
                     double *utraj, double *hint, int *status, int *ipm_iters, int cold);
void orc_rti_batch_mt_h(const orc_config *cfg, int B, const double *x0, const double *p, double *xtraj,
                        double *utraj, double *hint, int *status, int *ipm_iters, int cold, int nthreads);

/* ---- path front-end ---- */
/* getTestTrack (utilities.cpp:59-72): returns number of points written (<= cap) */
int orc_test_track(double *xy, int cap);
/* getTrackSection (PathTrackingSys.cpp:107-136) */
void orc_track_section(const double *xy, int n, double sx, double sy, int *start, int *end);
/* cubic polyfit by Householder QR (utilities.cpp:39-57) */
void orc_polyfit3(const double *xs, const double *ys, int n, double *coeffs);
/* calcCoeffs (PathTrackingSys.cpp:63-105): 4 rotations, min SSE; returns rotation */
double orc_calc_coeffs(const double *xy, int n, const double *state, double *coeffs);

#ifdef __cplusplus
}
#endif
#endif
