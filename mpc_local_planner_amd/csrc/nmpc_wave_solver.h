// nmpc_wave_solver.h -- one SQP-RTI solve of the path-tracking OCP by ONE workgroup of one
// 64-lane wavefront (N <= 63) or two (N <= 127) on gfx950.  This is the body of the HIP kernel in nmpc_hip.hip; it is
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
// Work mapping inside the wave (z ordering [u0,u1,X,Y,psi,v,delta,thr], augmented index 8 = "1")
//   * stage layout: lane k owns shooting node k (x_k,u_k, slacks, multipliers).  Linearisation
//     (ERK4 + forward sensitivities, Gauss-Newton cost) and all element-wise interior-point
//     work run here; wave-wide sums/maxima give the step lengths.
//   * factor sweep (serial in k): the Riccati recursion on the AUGMENTED matrices
//       Pt = [P p; p' .] (7x7),  BAt = [B A b] (6x9),  Ht = [H g; g' .] (9x9)
//     so that the gradient/gap vectors ride along as the 9th column.  Three phases per stage,
//     each a 6-7 term dot product per lane, exchanged through LDS:
//       phase 1  lane (m,j), 54 lanes : Mt[m][j] = Pt[m][:] . BAt[:][j]
//       phase 2  lane (i<=j), 44 lanes: Ht[i][j] = cost + BAt[:][i] . Mt[:][j]
//       phase 3  same lanes, i >= 2   : Schur complement on the 2x2 input block -> Pt, K, kff
//     Every lane addresses LDS through pointers set up once per sweep (stride 0 for constant
//     columns), so the loop body has no branches and no index arithmetic.
//   * vector sweeps (corrector backward, both forward passes): lane c = lane & 7 owns one row /
//     column; the 6-vector of the recursion stays in lanes of the wave and is consumed through
//     v_fmac_f64 with a DPP row broadcast (W::fmac_row), so the serial path has no LDS round
//     trip and no v_readlane.
//   The sweeps run on the first wave only; a second wave (N > 63) takes part in the stage layout.
//   Rows v,delta,thr of [B A] are constants of this model (a33, a35, b31, h) and columns X,Y are
//   unit vectors: they live in a small constant table, not per stage.
#pragma once

#ifndef NMPC_DEV
#define NMPC_DEV inline
#endif

#include <math.h>
#include <type_traits>

// Profiling aids (tools/prof_variants.py builds throw-away libraries with these; the shipped
// library defines neither): NMPC_PROF_FIXED_ITERS = run exactly that many interior-point
// iterations, NMPC_PROF_SKIP = bit mask of sweeps to leave out (1 factor, 2 backward, 4 forward)
// -- results are then meaningless, only the time difference is of interest.
#ifndef NMPC_PROF_SKIP
#define NMPC_PROF_SKIP 0
#endif

namespace nmpcw {

struct Problem {            // uniform over the launch
    int N;                  // horizon, N+1 <= 64 * waves per instance (<= 128)
    double dt;              // shooting interval
    double inv_wheelbase;   // 1/L
    int ipm_max_iter;
    double mu0, thr0, tau, tol_mu, tol_lin, tol_comp;
    int stall_iter;         // give up (status 4) when after this many iterations ... (0: never)
    double stall_lin;       // ... the infeasibility scale is still above this
    int sqp_max_iter;       // 1 = SQP-RTI (one QP per call)
    double sqp_tol;
    // active-set polish of the QP (nmpc_config: polish, polish_mu, ...): see "active-set solves" in solve_instance
    int polish;             // 0: interior point to the end
    double polish_mu;       // first attempt once the average complementarity is below this
    int polish_tries;       // primal-dual active-set corrections per attempt
    double polish_tol;      // KKT tolerance of acceptance
    int warm_tries;         // warm calls: corrections of the attempt on the iterate's own active set; < 0: never
    int warm_gate;          // ... tried when the previous QP moved at most this many pairs in or out of the set
    double warm_eps;        // iterate within this of a bound = guessed active
};

struct InstanceIO {
    const double *x0;       // [6] measured state (lbx0 = ubx0, AcadosSolver.cpp:15-26)
    const double *p;        // [6] c0,c1,c2,c3,pitch,v_ref (AcadosPathTracking.h:31-34)
    const double *W;        // [7] diagonal NLS weights
    const double *bnd;      // [8] lbu0,lbu1,lbx4,lbx5, ubu0,ubu1,ubx4,ubx5
    double *traj;           // [(N+1)*8] iterate, per node X,Y,psi,v,delta,thr,u0,u1 (warm start, no shift); the u0 slot
                            // of node N (no input there) carries the warm-start hint of the active-set polish
    double *ctrl;           // [6] u0[2], x1.delta, x1.throttle, cost, (status + 16*ipm iterations)
    int *status;            // acados numbering: 0 ok, 1 NaN, 2 max iter, 4 QP failure
    int *iters;             // interior-point iterations (sum over SQP iterations)
    int *sqp_iters;         // QPs solved
    double *cost;
    int cold;               // != 0: start from x_k = x0, u_k = 0 (AcadosSolver.cpp:78-92)
    int N;                  // horizon of THIS instance, 1 .. Problem::N (horizon sweeps inside one launch); 0 = Problem::N
};

// ---- LDS layout, in doubles.  [pseudo-stage -1 | stage 0 .. N-1 | exchange area]
// Everything read as a pair sits on a 16-byte boundary (ds_read_b128); the remaining reads are
// single ds_read_b64 -- the ds_read2_b64 form the compiler would otherwise merge them into costs
// 4x the LDS cycles of either (MI355X_MICROARCH.md, LDS table), and this kernel is LDS-bound.
// A stage holds six column groups of six doubles, one per variable column g = u0,u1,psi,v,delta,thr
// of [B A]:  (K0[g],K1[g]) | (BA[X][g],BA[Y][g]) | BA[psi][g] | unused.   The gain slots of the
// groups u0,u1 stay zero.  With this spacing a ROW of [B A] (offset G_XY + r) and a row of K
// (offset a) are both six loads at 6-double intervals from one base: the forward sweep fetches
// the coefficients of its row, whatever its role, with one address register.
constexpr int ST = 54;          // per stage
constexpr int GS = 6;           // group stride
constexpr int G_K = 0;          //   gain pair of the column (psi..thr); zero in the groups u0,u1
constexpr int G_XY = 2;         //   rows X,Y (pair), row psi at +2
constexpr int O_E = 36;         // (K0[X],K1[X]), (K0[Y],K1[Y])
constexpr int O_KFF = 40;       // (kff0,kff1)
constexpr int O_INV0 = 42;      // rows of -inv(Huu): (-i00,-i01) here, (-i01,-i11) at O_INV1
constexpr int O_D = 44;         // slot k: rb_k as [0,1,2,-,3,4,5,-] -> z_{k+1}[8] = du,dx -> dG_k[4], 0, 0
constexpr int O_INV1 = 52;
// The order of these tail blocks is chosen so that every per-stage read of the sweeps is free of
// LDS bank conflicts (tools/lds_conflicts.py): the gain pairs, the rows of -inv(Huu) and the start
// of O_D fall on different banks than the column groups they are read together with.  What is left
// are the lanes whose operand is a constant from a table outside the stage (unit columns, constant
// rows): they collide with the stage data at some stage alignments.
// input view of the factor sweep, overlaying the gain/inverse slots it writes afterwards
constexpr int I_GAM01 = 12, I_GAM23 = 18, I_Q0 = 24, I_Q2 = 30;   // (gam0,gam1) (gam2,gam3) (Q00,Q01) (Q02,-)
constexpr int I_G = 36;         // G[8]
// The constants above are the 54-double stage (Layout54).  Horizons at which its LDS image costs a
// resident instance per CU (compact_layout below: N = 91..102, three two-wave instances instead of four;
// N = 122..127, two instead of three; N = 50..55 and 59..63, six / five one-wave instances instead of
// seven / six) use the 48-double stage (Layout48): the same six column
// groups, but kff and the rows of -inv(Huu) live in the groups' spare slots (single doubles, six apart)
// and the gradient of the factor input is split between the O_E pairs and four spare slots.  Its stride
// of 96 dwords puts every other stage on the same banks (4-way conflicts on the stage-layout accesses);
// at these horizons the sweeps are > 99 % of the LDS traffic, and they address within one stage.
struct Layout54 {
    static constexpr int ST = 54, GS = 6, G_K = 0, G_XY = 2, O_E = 36, O_D = 44;
    static constexpr int KFF0 = 40, KFF1 = 41;                               // kff
    static constexpr int INV00 = 42, INV01 = 43, INV10 = 52, INV11 = 53;     // rows of -inv(Huu)
    static constexpr bool kPairOut = true;      // kff and each row of -inv(Huu) are 16-byte pairs
    static constexpr int I_GAM01 = 12, I_GAM23 = 18, I_Q0 = 24, I_Q2 = 30;
    static constexpr int g_off(int i) { return 36 + i; }                     // gradient entry i of the factor input
    static constexpr int LAM0 = 36, LAM1 = 38, LAM2 = 40, LAM3 = 42;         // SQP mode: bound multipliers between two QPs (pairs)
};
struct Layout48 {
    static constexpr int ST = 48, GS = 6, G_K = 0, G_XY = 2, O_E = 36, O_D = 40;
    static constexpr int KFF0 = 5, KFF1 = 11;
    static constexpr int INV00 = 17, INV01 = 23, INV10 = 29, INV11 = 35;
    static constexpr bool kPairOut = false;
    static constexpr int I_GAM01 = 12, I_GAM23 = 18, I_Q0 = 24, I_Q2 = 30;
    static constexpr int g_off(int i) { return i < 4 ? 36 + i : 5 + 6 * (i - 4); }
    static constexpr int LAM0 = 36, LAM1 = 38, LAM2 = 46, LAM3 = 30;         // O_E, the pad pair of O_D, gain pair of column thr
};
// pseudo-stage -1: its groups hold rows v,delta,thr of the six variable columns (constants) in
// the (X,Y,psi) positions; its O_D holds z_0
// exchange area after the stages
constexpr int X_P = 0;          // Pt rows [6][8]: P[m][0..5], p[m], pad
constexpr int X_M = 48;         // Mt col-major [9][6] (+ slack read by the idle lanes)
constexpr int X_HU = 116;       // (Ht[0][j], Ht[1][j]) j = 0..8, then the unit columns (1,0),(0,1)
constexpr int X_DUM = 138;      // dummy target for lanes with nothing to store
constexpr int X_UNIT = 142;     // (1,0,0,-),(0,1,0,-): rows X,Y,psi of columns X,Y
constexpr int X_ZERO = 150;     // 0,0,0,0
constexpr int X_FT = 154;       // 1,0,0,0,1,0: forward-sweep coefficients of dx_X, dx_Y for the state rows
constexpr int X_UNITB = 161;    // (1,0,0),-,(0,1,0): the unit columns again at ODD addresses for the backward sweep's
                                // 8-byte loads: the stage lanes read even, odd, even addresses there, the unit lanes
                                // odd, even, odd -- different banks at every stage alignment
#ifdef NMPC_WAVE_TIMELINE
constexpr int X_TL = 168;       // debug build (tools/wave_timeline.py): 24 per-phase clock accumulators of the sweeping wave
constexpr int X_SIZE = 168 + 24;
#else
constexpr int X_SIZE = 168;
#endif
// horizons beyond 63 use two waves per instance (lane = node needs 128 threads); their cross-wave
// reductions go through a small extra area owned by the wave policy
constexpr int X_POLICY = 8;
constexpr int waves_per_instance(int N) { return N + 1 <= 64 ? 1 : 2; }
constexpr int lds_doubles_st(int st, int N) { return (N + 1) * st + X_SIZE + (waves_per_instance(N) > 1 ? X_POLICY : 0); }
// instances that fit into a CU of the MI355X this layout choice is tuned for: 160 KB of LDS (allocated in
// 512-byte steps), 8 waves at up to 256 VGPRs (on another part the choice stays correct, only its tuning is off)
constexpr int instances_per_cu(int st, int N)
{
    const int bytes = (lds_doubles_st(st, N) * 8 + 511) / 512 * 512;
    const int cap = 8 / waves_per_instance(N);
    return 160 * 1024 / bytes < cap ? 160 * 1024 / bytes : cap;
}
// the 48-double stage where (and only where) it lets one more instance reside per CU
// ... and where that pays: a one-wave kernel gains nothing from an eighth instance per CU (measured at
// N = 44..48: 0.99-1.00x; 7 -> 8 waves only adds contention), 2-3 % from a seventh (N = 50..55), 6.5 % from
// a sixth (N = 59..63); two-wave kernels gain 16-30 % (profiles/r2/long_horizon_probe_compact_layout.txt)
constexpr bool compact_layout(int N)
{
    const int a = instances_per_cu(Layout54::ST, N), b = instances_per_cu(Layout48::ST, N);
    return b > a && (waves_per_instance(N) > 1 || a < 7);
}
constexpr int lds_doubles(int N) { return lds_doubles_st(compact_layout(N) ? Layout48::ST : Layout54::ST, N); }

// 16-byte pair access (ds_read_b128 / ds_write_b128) and a single load that is never merged
struct alignas(16) D2 { double x, y; };
NMPC_DEV D2 ld2(const double *p) { return *reinterpret_cast<const D2 *>(p); }
NMPC_DEV void st2(double *p, double a, double b) { *reinterpret_cast<D2 *>(p) = D2{a, b}; }
NMPC_DEV double ld1(const double *p) { return *p; }

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
    // (explicit fma: the shooting gap phi - x_{k+1} must round the same way in every instantiation of the
    // kernel, whatever the compiler would choose to contract around it -- mixed-horizon launches are
    // promised to be bit-identical to single-horizon ones)
    phi[0] = fma(h6, sX, x[0]);
    phi[1] = fma(h6, sY, x[1]);
    phi[2] = fma(h6, sP, x[2]);
    phi[3] = fma(h6, sV, x[3]);
    phi[4] = fma(h, u[0], x[4]);
    phi[5] = fma(h, u[1], x[5]);
    // columns: psi, v, delta, thr, u0, u1
    aX[0] = h6 * sXp; aX[1] = h6 * sXv; aX[2] = h6 * sXd; aX[3] = h6 * sXt; aX[4] = h6 * sX0; aX[5] = h6 * sX1;
    aY[0] = h6 * sYp; aY[1] = h6 * sYv; aY[2] = h6 * sYd; aY[3] = h6 * sYt; aY[4] = h6 * sY0; aY[5] = h6 * sY1;
    aP[0] = 1.0;      aP[1] = h6 * sPv; aP[2] = h6 * sPd; aP[3] = h6 * sPt; aP[4] = h6 * sP0; aP[5] = h6 * sP1;
    aV[0] = 0.0;      aV[1] = 1.0 + h6 * sVv; aV[2] = 0.0; aV[3] = h6 * sVt; aV[4] = 0.0;     aV[5] = h6 * sV1;
}

// entry of an 8-vector in z ordering that belongs to bounded variable q (u0,u1,delta,thr)
#define NMPC_ZQ(a, q) ((q) == 0 ? (a)[0] : (q) == 1 ? (a)[1] : (q) == 2 ? (a)[6] : (a)[7])
// variable-column index of z index j (u0,u1,-,-,psi,v,delta,thr) -> 0,1,-,-,2,3,4,5
#define NMPC_VCOL(j) ((j) < 2 ? (j) : (j) - 2)
// stage offset of the gain pair (K0[c],K1[c]), c = X,Y,psi,v,delta,thr
#define NMPC_KOFF(c) ((c) < 2 ? O_E + 2 * (c) : GS * (c) + G_K)
// state of pair q in a packed working set (0 free, 1 at its lower bound, 2 at its upper bound)
#define NMPC_WSQ(ws, q) (((ws) >> (2 * (q))) & 3)

template <class W>
NMPC_DEV void solve_instance(W &w, const Problem &pb, const InstanceIO &io)
{
    // stage layout of this instantiation (shadows the namespace-level constants of Layout54)
    using L = typename W::Layout;
    constexpr int ST = L::ST, GS = L::GS, G_K = L::G_K, G_XY = L::G_XY, O_E = L::O_E, O_D = L::O_D;
    constexpr int I_GAM01 = L::I_GAM01, I_GAM23 = L::I_GAM23, I_Q0 = L::I_Q0, I_Q2 = L::I_Q2;
    const int tid = w.tid();     // thread in the workgroup = shooting node in stage layout
    const int lane = w.lane();   // lane in the wave (the sweeps run on the first wave only)
    // the instance's own horizon: its nodes are threads 0..N, its stages the first N of the LDS image
    // (sized by the launch for Problem::N); everything below is written in terms of this N
    const int N = io.N > 0 ? io.N : pb.N;
    const double h = pb.dt;
    double *const ct = w.lds();                 // pseudo-stage -1 (constant tables, z_0)
    double *const stg = ct + ST;                // stage k at stg + k*ST
    double *const xa = ct + (N + 1) * ST;       // exchange area
    w.stamp_init(xa + X_SIZE - 24);            // (debug build NMPC_WAVE_TIMELINE: its accumulators are the area's last 24 slots)

    // ---------------------------------------------------------------- iterate (stage layout)
    const bool has_node = tid <= N;        // threads 0..N hold x_k
    const bool act = tid < N;              // threads 0..N-1 hold u_k and a shooting interval
    const bool actx = act && tid >= 1;     // state bounds exist on nodes 1..N-1 only
    const int node = has_node ? tid : N;
    double xb[6], ub_[2];                   // NLP iterate of this node
    {
        const double *t = io.traj + node * 8;
#pragma unroll
        for (int m = 0; m < 6; ++m) xb[m] = io.cold ? io.x0[m] : t[m];
        ub_[0] = (io.cold || !act) ? 0.0 : t[6];
        ub_[1] = (io.cold || !act) ? 0.0 : t[7];
    }
    // warm-start hint of the polish (0: none; 1 + n: the previous QP of this instance ended polished on an active
    // set that differed from the one its iterate suggested in n pairs); lives in the u0 slot of node N
    double hint = 0.0;
    if (pb.polish && pb.warm_tries >= 0 && !io.cold) hint = w.uniform(w.max(tid == N ? io.traj[N * 8 + 6] : 0.0));
    const double inv_mcount = w.uniform(w.rcp((double)(4 * N + 4 * (N - 1))));   // 1 / number of slack-multiplier pairs
    double lb[4], ub[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { lb[q] = io.bnd[q]; ub[q] = io.bnd[4 + q]; }
    // values that are the same on every lane are handed to the scalar registers (w.uniform): fp64
    // arithmetic leaves them in vector registers otherwise, and those are what this kernel is short of
    const double R00 = w.uniform(h * io.W[5]), R11 = w.uniform(h * io.W[6]), Q11 = w.uniform(h * io.W[0]),
                 Q22 = w.uniform(h * io.W[1]), Q33 = w.uniform(h * io.W[2]), Q44 = w.uniform(h * io.W[3]),
                 Q55 = w.uniform(h * io.W[4]);

    double *const last = stg + (N - 1) * ST;   // the backward sweeps start at stage N-1

    int total_ipm = 0, status = 0, sqp_it = 0;
    double cost_out = 0.0;
    bool tables_ready = false;
    double a33 = 0.0, a35 = 0.0, b31 = 0.0;   // constant entries of [B A] (row v)

    // SQP mode (W::kSqp, a compile-time property of the policy): the multipliers of the last QP, needed
    // for the KKT residuals at the next iterate, wait in the node's own stage of the LDS image --
    // costate pi_k where the adjoint sweep left it (O_D), bound multipliers in slots nothing writes
    // between two QPs (L::LAM0..3).  Held in registers they were live across
    // the integrator and cost 28 VGPRs of spills.

    for (sqp_it = 0;; ++sqp_it) {
        // ------------------------------------------------------------ linearise (stage layout)
        double rb[6], g[8], Q00, Q01, Q02;
        double kkt = 0.0;   // SQP mode: largest KKT residual of this node
        {
            double gp, gpc;
            sincos_bounded(io.p[4], &gp, &gpc);
            gp *= 9.81;
            double phi[6], aX[6], aY[6], aP[6], aV[6];
            if (W::kNeighbourLds) {
                // more than one wave: x_{k+1} comes through the gain slots of the column groups psi, v,
                // delta of the node's stage (written next by the factor sweep, not by the linearisation
                // of a neighbouring thread that runs ahead; node N: the same offsets of the exchange
                // area, idle at this point) -- one barrier for all six values.  Stores and barrier come
                // before the integrator so that its arithmetic and the shooting gap stay in one basic
                // block, as with the lane shuffles of the one-wave kernel: the compiler then contracts
                // multiply-adds the same way in both, and mixed-horizon launches stay bit-identical
                double *mine = stg + node * ST + 2 * GS + G_K;
                if (has_node) {         // (threads beyond node N hold a copy of x_N that is never stepped)
                    st2(mine, xb[0], xb[1]); st2(mine + GS, xb[2], xb[3]); st2(mine + 2 * GS, xb[4], xb[5]);
                }
                w.sync();
            }
            if (W::kSqp) {
                // SQP mode: the integrator sits in a loop, and everything in it that depends on the step
                // size alone (the whole v-row of the sensitivities) would be hoisted out of that loop and
                // then spilled -- the step size is made opaque per SQP iteration, so it is recomputed
                Problem pl = pb;
                pl.dt = w.rounded(pb.dt);
                rk4_sens(pl, xb, ub_, gp, phi, aX, aY, aP, aV);
            } else {
                rk4_sens(pb, xb, ub_, gp, phi, aX, aY, aP, aV);
            }
            double xn[6];                                     // x_{k+1}
            if (W::kNeighbourLds) {
                const double *nxt = stg + (tid < N ? tid + 1 : node) * ST + 2 * GS + G_K;
                const D2 n01 = ld2(nxt), n23 = ld2(nxt + GS), n45 = ld2(nxt + 2 * GS);
                xn[0] = n01.x; xn[1] = n01.y; xn[2] = n23.x; xn[3] = n23.y; xn[4] = n45.x; xn[5] = n45.y;
            } else {
#pragma unroll
                for (int m = 0; m < 6; ++m) xn[m] = w.shfl_down1(xb[m]);   // from the next lane
            }
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                rb[m] = w.rounded(act ? (w.rounded(phi[m]) - xn[m]) : 0.0);
                if (W::kSqp) kkt = fmax(kkt, fabs(rb[m]));                      // res_eq: shooting gaps
                if (W::kSqp && tid == 0) kkt = fmax(kkt, fabs(io.x0[m] - xb[m]));   //         and x_0 = x0
            }
            double Atp[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // SQP mode: [B A]' pi_{k+1} in z ordering
            double pik[6] = {0, 0, 0, 0, 0, 0}, lam_l[4] = {0, 0, 0, 0}, lam_u[4] = {0, 0, 0, 0};
            if (W::kSqp) {
                const bool have = sqp_it > 0;                 // no multipliers before the first QP
                double pn[6];                                 // pi_{k+1}: nodes 1..N-1 carry one, pi_N = 0
                {
                    const double *own = stg + (actx ? tid : 0) * ST + O_D;
                    const double *nx = stg + (tid + 1 < N ? tid + 1 : 0) * ST + O_D;
                    const bool hn = have && tid + 1 < N, ho = have && actx;
#pragma unroll
                    for (int m = 0; m < 6; ++m) { pn[m] = hn ? nx[m] : 0.0; pik[m] = ho ? own[m] : 0.0; }
                    const double *lm = stg + (act ? tid : 0) * ST;
                    const bool hl = have && act;
                    const D2 l01 = ld2(lm + L::LAM0), l23 = ld2(lm + L::LAM1), u01 = ld2(lm + L::LAM2), u23 = ld2(lm + L::LAM3);
                    lam_l[0] = hl ? l01.x : 0.0; lam_l[1] = hl ? l01.y : 0.0; lam_l[2] = hl ? l23.x : 0.0; lam_l[3] = hl ? l23.y : 0.0;
                    lam_u[0] = hl ? u01.x : 0.0; lam_u[1] = hl ? u01.y : 0.0; lam_u[2] = hl ? u23.x : 0.0; lam_u[3] = hl ? u23.y : 0.0;
                }
                Atp[0] = aX[4] * pn[0] + aY[4] * pn[1] + aP[4] * pn[2] + h * pn[4];
                Atp[1] = aX[5] * pn[0] + aY[5] * pn[1] + aP[5] * pn[2] + aV[5] * pn[3] + h * pn[5];
                Atp[2] = pn[0];
                Atp[3] = pn[1];
                Atp[4] = aX[0] * pn[0] + aY[0] * pn[1] + aP[0] * pn[2];
                Atp[5] = aX[1] * pn[0] + aY[1] * pn[1] + aP[1] * pn[2] + aV[1] * pn[3];
                Atp[6] = aX[2] * pn[0] + aY[2] * pn[1] + aP[2] * pn[2] + pn[4];
                Atp[7] = aX[3] * pn[0] + aY[3] * pn[1] + aP[3] * pn[2] + aV[3] * pn[3] + pn[5];
            }
            if (tid == 0) {   // fold the known dx_0 = x0 - x_0 into the first gap: b_0 += A_0 dx0
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
            if (act) {   // variable rows X,Y,psi of [B A] into the column groups u0,u1,psi,v,delta,thr
                double *c = stg + tid * ST + G_XY;
                st2(c + 0 * GS, aX[4], aY[4]); st2(c + 0 * GS + 2, aP[4], 0.0);
                st2(c + 1 * GS, aX[5], aY[5]); st2(c + 1 * GS + 2, aP[5], 0.0);
                st2(c + 2 * GS, aX[0], aY[0]); st2(c + 2 * GS + 2, aP[0], 0.0);
                st2(c + 3 * GS, aX[1], aY[1]); st2(c + 3 * GS + 2, aP[1], 0.0);
                st2(c + 4 * GS, aX[2], aY[2]); st2(c + 4 * GS + 2, aP[2], 0.0);
                st2(c + 5 * GS, aX[3], aY[3]); st2(c + 5 * GS + 2, aP[3], 0.0);
                // gain slots of the input columns: the forward sweep's input lanes read them as
                // their (zero) coefficients of du
                st2(c - G_XY + 0 * GS + G_K, 0.0, 0.0);
                st2(c - G_XY + 1 * GS + G_K, 0.0, 0.0);
            }
            if (!tables_ready) {   // constant rows (v,delta,thr) of [B A]: identical on every lane
                a33 = w.uniform(aV[1]); a35 = w.uniform(aV[3]); b31 = w.uniform(aV[5]);   // functions of dt only
                if (tid == 0) {
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
                    double *e = xa + X_HU + 18;     // unit columns: the lanes reading them get -inv(Huu)
                    e[0] = 1.0; e[1] = 0.0; e[2] = 0.0; e[3] = 1.0;
                }
                tables_ready = true;
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
            cost_out = w.uniform(w.sum(act ? cst : 0.0));
            if (W::kSqp) {
                // NLP KKT residuals at this iterate with the last QP's multipliers: acados' SQP
                // termination test (stationarity, equality, inequality, complementarity, each in the
                // infinity norm against one tolerance); stage N has neither cost nor bounds
                if (act) {
                    kkt = fmax(kkt, fabs(g[0] + Atp[0] - lam_l[0] + lam_u[0]));
                    kkt = fmax(kkt, fabs(g[1] + Atp[1] - lam_l[1] + lam_u[1]));
                }
                if (actx) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) kkt = fmax(kkt, fabs(g[2 + m] + Atp[2 + m] - pik[m]));
                    kkt = fmax(kkt, fabs(g[6] + Atp[6] - pik[4] - lam_l[2] + lam_u[2]));
                    kkt = fmax(kkt, fabs(g[7] + Atp[7] - pik[5] - lam_l[3] + lam_u[3]));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double v = q == 0 ? ub_[0] : q == 1 ? ub_[1] : q == 2 ? xb[4] : xb[5];
                    const bool on = q < 2 ? act : actx;
                    const double sl = v - lb[q], su = ub[q] - v;
                    if (on) kkt = fmax(kkt, fmax(fmax(-sl, -su), fmax(fabs(lam_l[q] * sl), fabs(lam_u[q] * su))));
                }
                const double res = w.max(kkt);
                if (sqp_it > 0 && res <= pb.sqp_tol) { status = 0; break; }
                if (sqp_it >= pb.sqp_max_iter) { status = 2; break; }
            }
        }

        // ------------------------------------------------------------ interior point (stage layout)
        // bounded variables of this node: u0,u1,delta,thr -> z index 0,1,6,7
        const double cur[4] = {ub_[0], ub_[1], xb[4], xb[5]};
        const bool mq[4] = {act, act, actx, actx};
        double tl[4], tu[4], ll[4], lu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double a = fmax(cur[q] - lb[q], pb.thr0), b = fmax(ub[q] - cur[q], pb.thr0);
            tl[q] = mq[q] ? a : 1.0;
            tu[q] = mq[q] ? b : 1.0;
            ll[q] = mq[q] ? pb.mu0 * w.rcp(a) : 0.0;
            lu[q] = mq[q] ? pb.mu0 * w.rcp(b) : 0.0;
        }
        double z[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // QP iterate of this node [du0,du1,dX..dthr]
        double lin_scale = 1.0;
        int it = 0, qp_status = 4;
        // ---- active-set solves ("polish").  The interior-point iteration identifies the active set long before
        // it has converged; from there the QP is finished exactly: the bounds classified active become equalities
        // (pins), the others are dropped, and ONE factor + forward sweep solves that equality-constrained QP from
        // the current point -- the pins are eliminated inside the factor stage (no penalties: a pinned input is
        // an affine function of the state, du_a = kap * dx_s + nu, substituted before the Schur complement).
        // One backward vector sweep with the full gradient at the result as right-hand side then delivers the
        // multipliers of the pins (its kff of a pinned input is minus the stationarity residual there).  The
        // result is accepted if it satisfies the KKT conditions of the QP; otherwise the working set is corrected
        // by the primal-dual active-set rule and the solve repeated from the same point, and after the allowed
        // number of corrections the interior-point iteration goes on where it was (its state is not touched).
        // Working sets are packed two bits per pair q (u0,u1,delta,thr): 0 free, 1 at lower, 2 at upper.
        int pol = 0, tries_left = 0, n_pol = 0;    // (uniform) this pass of the loop is an active-set solve; solves so far
        int wsp = 0, wsprev = -1, cls = 0;         // working set, last classification attempted, current classification
        int fixm = 0;                              // inputs of this stage fixed by a pin (bit 0: u0, bit 1: u1)
        double gscale = 1.0;                       // size of the gradient entries the multipliers are differences of
        bool polished = false;
        // a warm QP whose predecessor ended on (nearly) the set its iterate suggested: try the set this iterate
        // suggests first -- the bounded quantities that sit on a bound (the last polish left them there to round-off)
        if (pb.polish && hint >= 1.0 && hint <= 1.0 + pb.warm_gate) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mq[q] && lb[q] - cur[q] >= -pb.warm_eps) wsp |= 1 << (2 * q);
                else if (mq[q] && ub[q] - cur[q] <= pb.warm_eps) wsp |= 2 << (2 * q);
            }
            pol = 1;
            tries_left = pb.warm_tries;
        }
        w.sync();   // BA columns and constant tables visible
        w.stamp(0);      // (phase timeline of the debug build NMPC_WAVE_TIMELINE; no-ops otherwise)
        for (;;) {
            double mu = 0.0;
            if (!pol) {
            // ---- E-A: barrier system of this node
            double mu_loc = 0.0, c_loc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mu_loc += tl[q] * ll[q] + tu[q] * lu[q];
                // pairwise complementarity (masked pairs hold t = 1, lambda = 0)
                c_loc = fmax(c_loc, fmax(fmin(tl[q], ll[q]), fmin(tu[q], lu[q])));
            }
            // (a divide is an eleven-instruction dependent sequence on gfx950: reciprocal and multiply instead)
            mu = w.uniform(w.sum(mu_loc) * inv_mcount);
            // the pairwise test only decides once the averages have converged (uniform over the instance)
            const bool near_end = it > 0 && mu <= pb.tol_mu && lin_scale <= pb.tol_lin;
            const double cmax = near_end ? w.max(c_loc) : 1.0;
#ifndef NMPC_PROF_FIXED_ITERS      // (profiling variants that skip sweeps produce garbage, NaN included)
            if (!(mu == mu)) { qp_status = 1; break; }
#endif
#ifdef NMPC_PROF_FIXED_ITERS
            if (it >= NMPC_PROF_FIXED_ITERS) { qp_status = 0; break; }
#else
            if (near_end && cmax <= pb.tol_comp) { qp_status = 0; break; }
#endif
            if (it >= pb.ipm_max_iter) { qp_status = (mu <= 1e-8 && lin_scale <= 1e-8) ? 0 : 4; break; }
            // stalled: the step lengths have stayed tiny (infeasible or hopelessly started QP)
            if (pb.stall_iter > 0 && it >= pb.stall_iter && lin_scale > pb.stall_lin) { qp_status = 4; break; }
            if (pb.polish && it > 0 && mu <= pb.polish_mu) {
                // every pair classified by its slack against its multiplier as the last step predicted them at
                // full length (cls, from E-C); attempt when the classification is new
                if (w.max(cls != wsprev ? 1.0 : 0.0) != 0.0) { wsprev = cls; wsp = cls; pol = 1; tries_left = pb.polish_tries; }
            }
            }
            {
                double G[8];
                G[0] = R00 * z[0] + g[0];
                G[1] = R11 * z[1] + g[1];
                G[2] = Q00 * z[2] + Q01 * z[3] + Q02 * z[4] + g[2];
                G[3] = Q01 * z[2] + Q11 * z[3] + g[3];
                G[4] = Q02 * z[2] + Q22 * z[4] + g[4];
                G[5] = Q33 * z[5] + g[5];
                G[6] = Q44 * z[6] + g[6];
                G[7] = Q55 * z[7] + g[7];
                double gam[4];      // interior point: barrier diagonal; active-set solve: (kap0, nu0, kap1, nu1) of the stage
                if (!pol) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double wv = NMPC_ZQ(z, q);
                    const double rtl = mq[q] ? (wv - (lb[q] - cur[q]) - tl[q]) : 0.0;
                    const double rtu = mq[q] ? ((ub[q] - cur[q]) - wv - tu[q]) : 0.0;
                    const double rl = ll[q] * w.rcp(tl[q]), ru = lu[q] * w.rcp(tu[q]);
                    gam[q] = rl + ru;
                    const double gq = rl * rtl - ru * rtu;
                    if (q == 0) G[0] += gq;
                    if (q == 1) G[1] += gq;
                    if (q == 2) G[6] += gq;
                    if (q == 3) G[7] += gq;
                }
                } else {
                    // pins of stage `tid`: its own input pins, and the state pins of node tid + 1, which fix the
                    // input that integrates into that state (rows delta, thr of [B A] are exact integrators:
                    // d delta_{k+1} = d delta_k + h du0_k + rb, the same for thr and u1).  Targets are increments
                    // from the current point z.  An input pinned together with its state of the next node would
                    // leave the stage with dependent rows: the state's pin is kept, the input freed (if its bound
                    // then fails, the correction after the solve says so).
                    constexpr double FREE = 1e300;
                    double tq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int s = NMPC_WSQ(wsp, q);
                        tq[q] = s == 0 ? FREE : ((s == 1 ? lb[q] : ub[q]) - cur[q]) - NMPC_ZQ(z, q);
                    }
                    double tn0, tn1;     // targets of delta, thr at the next node
                    if (W::kNeighbourLds) {
                        if (act) st2(stg + tid * ST + I_GAM23, tq[2], tq[3]);
                        w.sync();
                        const D2 nx = ld2(stg + (tid + 1 < N ? tid + 1 : 0) * ST + I_GAM23);
                        tn0 = nx.x; tn1 = nx.y;
                        w.sync();
                    } else {
                        tn0 = w.shfl_down1(tq[2]); tn1 = w.shfl_down1(tq[3]);
                    }
                    if (!(tid + 1 < N)) { tn0 = FREE; tn1 = FREE; }
                    const double ih = w.rcp(h);
                    if (tn0 < 1e299) { gam[0] = -ih; gam[1] = (tn0 - rb[4]) * ih; wsp &= ~3; }
                    else if (tq[0] < 1e299) { gam[0] = 0.0; gam[1] = tq[0]; }
                    else { gam[0] = 1.0; gam[1] = 0.0; }
                    if (tn1 < 1e299) { gam[2] = -ih; gam[3] = (tn1 - rb[5]) * ih; wsp &= ~12; }
                    else if (tq[1] < 1e299) { gam[2] = 0.0; gam[3] = tq[1]; }
                    else { gam[2] = 1.0; gam[3] = 0.0; }
                    fixm = (gam[0] <= 0.5 ? 1 : 0) | (gam[2] <= 0.5 ? 2 : 0);
                }
                if (act) {
                    double *sl = stg + tid * ST;
                    // 16-byte stores: with the stage stride of 100 dwords they are bank-conflict free
                    // across lanes (8-byte accesses are 2-way)
                    st2(sl + I_GAM01, gam[0], gam[1]); st2(sl + I_GAM23, gam[2], gam[3]);
                    st2(sl + I_Q0, Q00, Q01);          st2(sl + I_Q2, Q02, 0.0);
                    if (L::kPairOut) {
#pragma unroll
                        for (int m = 0; m < 8; m += 2) st2(sl + L::g_off(m), G[m], G[m + 1]);
                    } else {
                        st2(sl + L::g_off(0), G[0], G[1]); st2(sl + L::g_off(2), G[2], G[3]);
#pragma unroll
                        for (int m = 4; m < 8; ++m) sl[L::g_off(m)] = G[m];
                    }
                    st2(sl + O_D + 0, rb[0], rb[1]); st2(sl + O_D + 2, rb[2], 0.0);
                    st2(sl + O_D + 4, rb[3], rb[4]); st2(sl + O_D + 6, rb[5], 0.0);
                }
            }
            // terminal cost-to-go: none (no *_e cost or bound in mpcPathTracking.py)
            if (tid < 48) xa[X_P + tid] = 0.0;
            w.sync();
            w.stamp(pol ? 8 : 1);

            // ---- factor sweep, k = N-1 .. 0 (augmented Riccati recursion), first wave only
            w.sweep_priority(true);
            if (w.sweep_wave()) {
                // per-lane LDS pointers of this sweep, recomputed here (a few dozen integer ops per
                // interior-point iteration) instead of living in ~45 registers through the whole solve;
                // strides are in doubles per stage: ST for per-stage data, 0 for constants
                const int fl = w.opaque(lane);
                // phase 1: lane (m,j) = (fl/9, fl%9), lanes >= 54 idle (dummy store)
                const int f1m = fl < 54 ? fl / 9 : 5, f1j = fl < 54 ? fl % 9 : 8;
                // phases 2,3: triangular map lane -> (i <= j), i in 0..7, j in 0..8; lanes 44,45 read the
                // unit columns 9,10 of the input rows and so produce the two rows of -inv(Huu)
                int f2i = 7, f2j = 8;
                {
                    const int start[9] = {0, 9, 17, 24, 30, 35, 39, 42, 44};
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (fl >= start[i] && fl < start[i + 1]) { f2i = i; f2j = i + (fl - start[i]); }
                }
                const bool f2on = fl < 44;
                if (fl == 44 || fl == 45) f2j = 9 + (fl - 44);
                // column j of BAt for phase 1 (top = rows X,Y,psi, bot = rows v,delta,thr)
                const double *tj, *bj;
                int sTopJ, sBotJ;
                if (f1j == 8) { tj = last + O_D; bj = last + O_D + 4; sTopJ = ST; sBotJ = ST; }
                else if (f1j == 2 || f1j == 3) { tj = xa + X_UNIT + 4 * (f1j - 2); bj = xa + X_ZERO; sTopJ = 0; sBotJ = 0; }
                else { tj = last + GS * NMPC_VCOL(f1j) + G_XY; bj = ct + GS * NMPC_VCOL(f1j) + G_XY; sTopJ = ST; sBotJ = 0; }
                // column i of BAt for phase 2 (i < 8)
                const double *ti, *pBotI;
                int sTopI;
                if (f2i == 2 || f2i == 3) { ti = xa + X_UNIT + 4 * (f2i - 2); pBotI = xa + X_ZERO; sTopI = 0; }
                else { ti = last + GS * NMPC_VCOL(f2i) + G_XY; pBotI = ct + GS * NMPC_VCOL(f2i) + G_XY; sTopI = ST; }
                const double e8 = (f1j == 8) ? 1.0 : 0.0;
                const double *const pProw = xa + X_P + f1m * 8;
                double *const pMw = fl < 54 ? xa + X_M + f1j * 6 + f1m : xa + X_DUM;
                const double *const pMr = xa + X_M + f2j * 6;
                // cost part of Ht[i][j]: a constant plus (at most) one per-stage table entry
                double hconst = 0.0;
                const double *tb = xa + X_ZERO;
                int sTab = 0;
                {
                    int off = -1;
                    if (f2j == 8) {                                          // gradient column
#pragma unroll
                        for (int i = 0; i < 8; ++i) if (f2i == i) off = L::g_off(i);
                    }
                    else if (f2i == f2j) {
                        if (f2i == 0) { hconst = R00; off = I_GAM01; }
                        if (f2i == 1) { hconst = R11; off = I_GAM01 + 1; }
                        if (f2i == 2) off = I_Q0;
                        if (f2i == 3) hconst = Q11;
                        if (f2i == 4) hconst = Q22;
                        if (f2i == 5) hconst = Q33;
                        if (f2i == 6) { hconst = Q44; off = I_GAM23; }
                        if (f2i == 7) { hconst = Q55; off = I_GAM23 + 1; }
                    } else if (f2i == 2 && f2j == 3) off = I_Q0 + 1;
                    else if (f2i == 2 && f2j == 4) off = I_Q2;
                    // (an active-set solve has no barrier diagonal: those slots carry the stage's pins instead)
                    const bool is_gam = off == I_GAM01 || off == I_GAM01 + 1 || off == I_GAM23 || off == I_GAM23 + 1;
                    if (off >= 0 && !(pol && is_gam)) { tb = last + off; sTab = ST; }
                }
                // active-set solve: the pins of the stage, (kap0, nu0) and (kap1, nu1): kap = 1 free input, else
                // du_a = kap * dx_s + nu (s = delta for u0, thr for u1); selectors of this lane's row i and column j
                const double *tpin = last + I_GAM01;
                const int pij = w.opaque(f2i | (f2j << 4));     // (row, column) of this lane, one register for the stage
                // phase 2 store of the two input rows (pairs (Ht[0][j], Ht[1][j])), phase 3 loads/stores
                double *const pHw = (f2on && f2i < 2) ? xa + X_HU + 2 * f2j + f2i : xa + X_DUM;
                const double *const pHj = xa + X_HU + 2 * f2j;
                const double *const pHi = xa + X_HU + 2 * f2i;
                // (kw, kw2): where this lane's pair (K0, K1) goes -- gains of a column, kff, a row of -inv(Huu);
                // one 16-byte store where the layout keeps them side by side, two 8-byte stores otherwise
                double *pPw1 = xa + X_DUM, *pPw2 = xa + X_DUM, *kw = xa + X_DUM, *kw2 = xa + X_DUM + 1;
                int sKw = 0;
                if (f2on && f2i >= 2) {
                    pPw1 = xa + X_P + (f2i - 2) * 8 + (f2j - 2);
                    if (f2j <= 7) pPw2 = xa + X_P + (f2j - 2) * 8 + (f2i - 2);
                    if (f2i == 2) {
                        if (f2j < 8) { kw = last + NMPC_KOFF(f2j - 2); kw2 = kw + 1; }
                        else { kw = last + L::KFF0; kw2 = last + L::KFF1; }
                        sKw = ST;
                    }
                }
                if (fl == 44) { kw = last + L::INV00; kw2 = last + L::INV01; sKw = ST; }
                if (fl == 45) { kw = last + L::INV10; kw2 = last + L::INV11; sKw = ST; }
                // rows v,delta,thr of column i are constants: loaded once per sweep
                const D2 v01 = ld2(pBotI);
                const double v2 = pBotI[2];
                // stage data (independent of the recursion) is fetched one stage ahead, so that the
                // loads that ARE on the serial path (Pt row, Mt column, input rows) never queue behind it;
                // two register sets take turns (loop body written out twice: no register moves)
                struct StageData { D2 t01, b01, u01; double t2, b2, u2, tbv; D2 pin0, pin1; };
                StageData sdA, sdB;
                sdA.t01 = ld2(tj); sdA.b01 = ld2(bj); sdA.u01 = ld2(ti);
                sdA.t2 = tj[2]; sdA.b2 = bj[2]; sdA.u2 = ti[2]; sdA.tbv = *tb;
                // (POL: std::true_type for an active-set solve -- a second instantiation of the stage, so that the
                // interior-point iterations carry none of the pin arithmetic.  In the one-wave kernels that
                // instantiation runs with ONE register set, c and n the same object: each operand group is
                // refilled for the next stage right after its last use in this one, still a phase ahead of its
                // next use, and the stage's pins are loaded at its start for its own phase 3 -- two sets plus
                // pins in flight do not fit in 256 registers beside the interior point's state, which must
                // survive a rejected attempt.  The two-wave kernels keep two sets: their sweeping wave has a SIMD
                // to itself, nothing hides a later load there, and the single set measured -5 % on config 4)
                constexpr bool kOneSet = !W::kNeighbourLds;
                if (pol && !kOneSet) { sdA.pin0 = ld2(tpin); sdA.pin1 = ld2(tpin + (I_GAM23 - I_GAM01)); }
                auto factor_stage = [&](auto POL, const StageData &c, StageData &n) {
                    constexpr bool kPol = decltype(POL)::value;
                    const D2 r01 = ld2(pProw), r23 = ld2(pProw + 2), r45 = ld2(pProw + 4);
                    const double r6 = pProw[6];
                    D2 pin0 = {1.0, 0.0}, pin1 = {1.0, 0.0};
                    if (kPol && kOneSet) { pin0 = ld2(tpin); pin1 = ld2(tpin + (I_GAM23 - I_GAM01)); tpin -= ST; }
                    if (kPol && !kOneSet) { pin0 = c.pin0; pin1 = c.pin1; }
                    w.sched_fence();
                    // phase 1: Mt[m][j] = Pt[m][0..5] . BAt[0..5][j] + p[m] [j == 8]
                    double accA = r6 * e8, accB = r01.y * c.t01.y;
                    accA = fma(r01.x, c.t01.x, accA); accB = fma(r23.y, c.b01.x, accB);
                    accA = fma(r23.x, c.t2, accA);    accB = fma(r45.y, c.b2, accB);
                    accA = fma(r45.x, c.b01.y, accA);
                    *pMw = accA + accB;
                    w.wsync();
                    // phase 2: Ht[i][j] = cost[i][j] + BAt[0..5][i] . Mt[0..5][j]
                    const D2 m01 = ld2(pMr), m23 = ld2(pMr + 2), m45 = ld2(pMr + 4);
                    // data of stage k-1 (at k = 0 this reads the pseudo-stage: valid memory, not used)
                    tj -= sTopJ; bj -= sBotJ; ti -= sTopI; tb -= sTab;
                    n.t01 = ld2(tj); n.b01 = ld2(bj);
                    n.t2 = tj[2]; n.b2 = bj[2];
                    if (!(kPol && kOneSet)) { n.u01 = ld2(ti); n.u2 = ti[2]; n.tbv = *tb; }
                    if (kPol && !kOneSet) { tpin -= ST; n.pin0 = ld2(tpin); n.pin1 = ld2(tpin + (I_GAM23 - I_GAM01)); }
                    w.sched_fence();
                    double HvA = fma(c.u01.x, m01.x, hconst + c.tbv), HvB = c.u01.y * m01.y;
                    HvA = fma(c.u2, m23.x, HvA);  HvB = fma(v01.x, m23.y, HvB);
                    HvA = fma(v01.y, m45.x, HvA); HvB = fma(v2, m45.y, HvB);
                    const double Hv = HvA + HvB;
                    *pHw = Hv;
                    if (kPol && kOneSet) { w.sched_fence(); n.u01 = ld2(ti); n.u2 = ti[2]; n.tbv = *tb; }
                    // the 2x2 input block straight from its lanes (0: H00, 1: H01, 9: H11): its inverse is
                    // worked out while the input rows make their way through LDS
                    const double H00 = w.readlane(Hv, 0), H01 = w.readlane(Hv, 1), H11 = w.readlane(Hv, 9);
                    w.wsync();
                    // phase 3: eliminate the inputs
                    const D2 hj = ld2(pHj), hi = ld2(pHi);
                    w.sched_fence();
                    double Hvp = Hv, hjx = hj.x, hjy = hj.y, hix = hi.x, hiy = hi.y, h00 = H00, h01 = H01, h11 = H11;
                    double t0j = 0.0, t1j = 0.0;
                    // (a stage without pins -- most of them -- is the plain stage: the pins are the same on every
                    // lane, so this is a branch of the whole wave, and skipping adds nothing but exact zeros)
                    if (kPol && w.uniform_flag(pin0.x <= 0.5 || pin1.x <= 0.5)) {
                        // substitute the pinned inputs du_a = t_a . (dx, 1) into the stage's quadratic, then treat
                        // them as decoupled dummies (unit diagonal, zero row) in the elimination below
                        const bool f0 = pin0.x <= 0.5, f1 = pin1.x <= 0.5;
                        const double k0 = f0 ? pin0.x : 0.0, n0 = f0 ? pin0.y : 0.0;
                        const double k1 = f1 ? pin1.x : 0.0, n1 = f1 ? pin1.y : 0.0;
                        const int pi_ = pij & 15, pj_ = pij >> 4;
                        const bool jcol = pj_ <= 8;     // (lanes 44,45 carry unit columns, not columns of Ht)
                        t0j = pj_ == 6 ? k0 : pj_ == 8 ? n0 : 0.0;
                        t1j = pj_ == 7 ? k1 : pj_ == 8 ? n1 : 0.0;
                        const double t0i = pi_ == 6 ? k0 : 0.0, t1i = pi_ == 7 ? k1 : 0.0;
                        Hvp = Hv + t0i * hj.x + t0j * hi.x + t0i * t0j * H00 + t1i * hj.y + t1j * hi.y + t1i * t1j * H11 +
                              (t0i * t1j + t1i * t0j) * H01;
                        hjx = (f0 && jcol) ? 0.0 : hj.x + t1j * H01;
                        hjy = (f1 && jcol) ? 0.0 : hj.y + t0j * H01;
                        hix = f0 ? 0.0 : hi.x + t1i * H01;
                        hiy = f1 ? 0.0 : hi.y + t0i * H01;
                        h00 = f0 ? 1.0 : H00;
                        h11 = f1 ? 1.0 : H11;
                        h01 = (f0 || f1) ? 0.0 : H01;
                    }
                    const double rdet = w.rcp(h00 * h11 - h01 * h01);
                    // rows of -inv(Huu) (the signs ride on the multiplies): K = -inv(Huu) Hu
                    const double n00 = -h11 * rdet, n01 = h01 * rdet, n11 = -h00 * rdet;
                    double K0 = fma(n00, hjx, n01 * hjy), K1 = fma(n01, hjx, n11 * hjy);
                    const double Pn = Hvp + hix * K0 + hiy * K1;
                    if (kPol) { K0 += t0j; K1 += t1j; }      // the gains as the forward sweep applies them
                    *pPw1 = Pn;
                    *pPw2 = Pn;
                    if (L::kPairOut) st2(kw, K0, K1);
                    else { *kw = K0; *kw2 = K1; }
                    w.wsync();
                    kw -= sKw;
                    if (!L::kPairOut) kw2 -= sKw;
                };
                if (!pol) {
#pragma unroll 1
                    for (int k = (NMPC_PROF_SKIP & 1) ? -1 : N - 1; k >= 0; k -= 2) {
                        factor_stage(std::false_type(), sdA, sdB);
                        if (k == 0) break;
                        factor_stage(std::false_type(), sdB, sdA);
                    }
                } else {
                    if (kOneSet) {
#pragma unroll 1
                        for (int k = N - 1; k >= 0; --k) factor_stage(std::true_type(), sdA, sdA);
                    } else {
#pragma unroll 1
                        for (int k = N - 1; k >= 0; k -= 2) {
                            factor_stage(std::true_type(), sdA, sdB);
                            if (k == 0) break;
                            factor_stage(std::true_type(), sdB, sdA);
                        }
                    }
                }
            }
            w.stamp(pol ? 9 : 2);

            // ---- pass 0: forward sweep with the affine right-hand side, then the corrector
            //      pass 1: backward + forward sweep of the corrector increment (homogeneous dynamics)
            double dzs[8];        // affine step of this node (active-set solve: its step)
            double smu = 0.0, csc = 0.0, mu_aff = 0.0;
#pragma unroll 1
            // (an active-set solve has three passes: forward sweep of its step; backward sweep for the multipliers
            // of its pins, which is also the first half of one step of iterative refinement -- the gradient at the
            // new point is the residual of the solve; forward sweep of that correction)
            for (int pass = 0; pass < (pol ? 3 : 2); ++pass) {
                if (w.sweep_wave() && pass == 1) {
                    // backward vector sweep (corrector right-hand side, homogeneous dynamics).  Lane
                    // vl = lane & 7 owns z component vl; the result of a stage (lanes 0,1: kff, lanes
                    // 2..7: the costate p) feeds the next stage through fmac_row, the fp64 multiply-add
                    // whose first factor is another lane of the same 16-lane row: the serial path has
                    // neither an LDS round trip nor a v_readlane.
                    const int bl = w.opaque(lane);
                    const int vl = bl & 7;
                    const double *tv;                 // column vl of [B A], rows X,Y,psi
                    int sTopV;
                    if (vl == 2 || vl == 3) { tv = xa + X_UNITB + 4 * (vl - 2); sTopV = 0; }
                    else { tv = last + GS * NMPC_VCOL(vl) + G_XY; sTopV = ST; }
                    // corrector gradient entry of component vl; zero for the other components, read from
                    // the zero the stage keeps next to it (this operand is also the accumulator of the
                    // stage's multiply-add chain, so it is loaded afresh every stage by all lanes -- from
                    // inside the stage, hence without bank conflicts)
                    // (an active-set solve runs this sweep with the full gradient as right-hand side -- components
                    // X..v in slots 4..7 -- to recover the multipliers of its pins; the corrector leaves zeros there)
                    const double *dg = last + O_D + (vl < 2 ? vl : vl >= 6 ? vl - 4 : vl + 2);
                    const int sDgV = ST;
                    // res = keep * hc + ab.x * hc[0] + ab.y * hc[1]: lanes >= 2 form p_new with their gain
                    // pair (K0,K1); lanes 0,1 form kff = -(inv hu) with row vl of -inv
                    const double *ab = vl >= 2 ? last + NMPC_KOFF(vl - 2) : last + (vl == 0 ? L::INV00 : L::INV10);
                    const double *ab2 = vl >= 2 ? ab + 1 : last + (vl == 0 ? L::INV01 : L::INV11);   // (second entry: only where not adjacent)
                    const double keep = vl >= 2 ? 1.0 : 0.0;
                    double *kfw = bl < 2 ? last + (bl == 0 ? L::KFF0 : L::KFF1) : xa + X_DUM;
                    const int sKfw = bl < 2 ? ST : 0;
                    // rows v,delta,thr of column vl
                    const double cb0 = vl == 1 ? b31 : vl == 5 ? a33 : vl == 7 ? a35 : 0.0;
                    const double cb1 = vl == 0 ? h : vl == 6 ? 1.0 : 0.0;
                    const double cb2 = vl == 1 ? h : vl == 7 ? 1.0 : 0.0;
                    // stage data is fetched one stage ahead (two register sets, loop unrolled twice): the
                    // LDS latency then hides behind the multiply-add chain of the current stage
                    double res = 0.0;
                    double da[6], db[6];      // column (X,Y,psi), gradient entry, gain pair
                    if (L::kPairOut) {
                        w.bwd_fetch(tv, dg, ab, da);
#pragma unroll 1
                        for (int k = (NMPC_PROF_SKIP & 2) ? -1 : N - 1; k >= 0; k -= 2) {
                            tv -= sTopV; dg -= sDgV; ab -= ST;
                            w.bwd_stage(da, db, tv, dg, ab, kfw, cb0, cb1, cb2, keep, res);
                            kfw -= sKfw;
                            if (k == 0) break;
                            tv -= sTopV; dg -= sDgV; ab -= ST;
                            w.bwd_stage(db, da, tv, dg, ab, kfw, cb0, cb1, cb2, keep, res);
                            kfw -= sKfw;
                        }
                    } else {
                        w.bwd_fetch2(tv, dg, ab, ab2, da);
#pragma unroll 1
                        for (int k = (NMPC_PROF_SKIP & 2) ? -1 : N - 1; k >= 0; k -= 2) {
                            tv -= sTopV; dg -= sDgV; ab -= ST; ab2 -= ST;
                            w.bwd_stage2(da, db, tv, dg, ab, ab2, kfw, cb0, cb1, cb2, keep, res);
                            kfw -= sKfw;
                            if (k == 0) break;
                            tv -= sTopV; dg -= sDgV; ab -= ST; ab2 -= ST;
                            w.bwd_stage2(db, da, tv, dg, ab, ab2, kfw, cb0, cb1, cb2, keep, res);
                            kfw -= sKfw;
                        }
                    }
                    w.sweep_drain();
                    w.wsync();
                    w.stamp(pol ? 12 : 5);
                }
                // forward sweep.  Lanes 0..5 of every 16-lane row own a row of dx_{k+1}, lanes 6,7 own
                // du_k[0], du_k[1] (lanes 8..15 repeat 0..7); dx_k sits in lanes 0..5 and reaches the
                // others through fmac_row.  Level 1: every lane starts from r_b / kff and adds its six
                // coefficients (row of A, or row of K) times dx_k; level 2: the state rows add their
                // B entries times du (lanes 6,7 read zeros there and keep du).  z_k is stored in slot
                // k-1 (its r_b has been consumed).
                if (w.sweep_wave() && !(pol && pass == 1)) {
                    const int fl = w.opaque(lane);
                    const int vl = fl & 7;
                    // six coefficients at group spacing: columns u0,u1 (level 2) and psi,v,delta,thr
                    const double *p1;
                    int s1;
                    if (vl < 3) { p1 = stg + G_XY + vl; s1 = ST; }                 // rows X,Y,psi of [B A]
                    else if (vl < 6) { p1 = ct + G_XY + (vl - 3); s1 = 0; }        // rows v,delta,thr: constants
                    else { p1 = stg + G_K + (vl - 6); s1 = ST; }                   // K_a[psi..thr], zeros for u0,u1
                    // coefficients of dx_X, dx_Y: unit columns for the state rows, K_a[X], K_a[Y] for the inputs
                    const double *p3;
                    int s3;
                    if (vl < 6) { p3 = xa + X_FT + (vl == 0 ? 0 : vl == 1 ? 2 : 1); s3 = 0; }
                    else { p3 = stg + O_E + (vl - 6); s3 = ST; }
                    // start value: r_b (affine pass only) or kff
                    const double *p2;
                    int s2;
                    if (vl >= 6) { p2 = stg + (vl == 6 ? L::KFF0 : L::KFF1); s2 = ST; }
                    else if (pass == 0) { p2 = stg + O_D + (vl < 3 ? vl : vl + 1); s2 = ST; }
                    else { p2 = stg + O_D + 4; s2 = ST; }     // the zero next to the corrector rhs (see E-B)
                    double *sw = xa + X_DUM;          // dx_{k+1}[r] -> slot k, du_k[a] -> slot k-1
                    int sSw = 0;
                    if (fl < 6) { sw = stg + O_D + 2 + fl; sSw = ST; }
                    else if (fl < 8) { sw = ct + O_D + (fl - 6); sSw = ST; }
                    // coefficients are fetched one stage ahead, as in the backward sweep
                    double val = 0.0;
                    double fa[9], fb[9];      // six group-spaced coefficients, two for dx_X, dx_Y, start value
                    w.fwd_fetch(p1, p3, p2, fa);
#pragma unroll 1
                    for (int k = (NMPC_PROF_SKIP & 4) ? N : 0; k < N; k += 2) {
                        p1 += s1; p2 += s2; p3 += s3;
                        w.fwd_stage(fa, fb, p1, p3, p2, sw, val);
                        sw += sSw;
                        if (k + 1 == N) break;
                        p1 += s1; p2 += s2; p3 += s3;
                        w.fwd_stage(fb, fa, p1, p3, p2, sw, val);
                        sw += sSw;
                    }
                    w.sweep_drain();
                    if (lane == 0) {                // z_0 = (du_0, 0): dx_0 is folded into r_b
#pragma unroll
                        for (int m = 0; m < 6; ++m) ct[O_D + 2 + m] = 0.0;
                    }
                }
                w.sweep_priority(false);
                w.sync();
                if (!(pol && pass == 1)) w.stamp(pol ? (pass == 0 ? 10 : 14) : (pass == 0 ? 3 : 6));
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
                if (pol) {
                    if (pass == 0) {
                        // the solve's step from z; right-hand side of the multiplier sweep: the full gradient there
#pragma unroll
                        for (int m = 0; m < 8; ++m) dzs[m] = dzl[m];
                        double v[8];
#pragma unroll
                        for (int m = 0; m < 8; ++m) v[m] = z[m] + dzl[m];
                        const double G0 = R00 * v[0] + g[0], G1 = R11 * v[1] + g[1];
                        const double G2 = Q00 * v[2] + Q01 * v[3] + Q02 * v[4] + g[2], G3 = Q01 * v[2] + Q11 * v[3] + g[3];
                        const double G4 = Q02 * v[2] + Q22 * v[4] + g[4], G5 = Q33 * v[5] + g[5];
                        const double G6 = Q44 * v[6] + g[6], G7 = Q55 * v[7] + g[7];
                        // (this reduction also orders every node's pick-up before the stores below)
                        const double gl = fmax(fmax(fmax(fabs(G0), fabs(G1)), fmax(fabs(G2), fabs(G3))),
                                               fmax(fmax(fabs(G4), fabs(G5)), fmax(fabs(G6), fabs(G7))));
                        gscale = w.uniform(fmax(1.0, w.max(act ? gl : 0.0)));
                        if (act) {
                            double *sl = stg + tid * ST + O_D;
                            st2(sl, G0, G1); st2(sl + 2, G6, G7); st2(sl + 4, G2, G3); st2(sl + 6, G4, G5);
                        }
                        w.sync();
                        w.sweep_priority(true);
                    } else if (pass == 1) {
                        // multipliers of the pins: the sweep left minus the stationarity residual of a pinned input
                        // in the stage's kff slots -- that is the signed multiplier of an input bound, and h times
                        // that of the state bound (next node) the input was fixed by
                        const double *own = stg + (act ? tid : 0) * ST, *prv = ct + (actx ? tid : 1) * ST;
                        const double k0 = own[L::KFF0], k1 = own[L::KFF1];
                        const double smq[4] = {k0, k1, prv[L::KFF0] * w.rcp(h), prv[L::KFF1] * w.rcp(h)};
                        // KKT conditions of the QP at z + dzs on the working set; its correction where they fail
                        const double tol_d = pb.polish_tol * gscale;
                        int bad = 0, wsn = wsp;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int s = NMPC_WSQ(wsp, q);
                            const double vq = NMPC_ZQ(z, q) + NMPC_ZQ(dzs, q);
                            const double lo = lb[q] - cur[q], up = ub[q] - cur[q];
                            const bool below = !(vq >= lo - pb.polish_tol), above = !(vq <= up + pb.polish_tol);
                            if (mq[q] && s == 0 && (below || above)) { bad = 1; wsn |= (below ? 1 : 2) << (2 * q); }
                            if (mq[q] && s == 1 && !(smq[q] <= tol_d)) { bad = 1; wsn &= ~(3 << (2 * q)); }
                            if (mq[q] && s == 2 && !(smq[q] >= -tol_d)) { bad = 1; wsn &= ~(3 << (2 * q)); }
                        }
                        ++n_pol;
                        w.tick(18);
                        // (the reduction also orders every node's reads of its neighbour's slots before the stores below)
                        if (w.max(bad ? 1.0 : 0.0) == 0.0) {
                            // the vertex: pinned quantities sit on their bounds to round-off, complementarity is
                            // exact.  One step of iterative refinement follows (pass 2): the backward sweep just
                            // run is its first half -- the gradient at the new point is the residual of the
                            // solve -- and a forward sweep in which a fixed input stays where it is (no
                            // feed-forward) and the state rows start from zero (homogeneous dynamics) the second.
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int s = NMPC_WSQ(wsp, q);
                                ll[q] = (mq[q] && s == 1) ? fmax(-smq[q], 0.0) : 0.0;
                                lu[q] = (mq[q] && s == 2) ? fmax(smq[q], 0.0) : 0.0;
                            }
                            polished = true;
                            if (act) {
                                double *sl = stg + tid * ST;
                                if (L::kPairOut) st2(sl + L::KFF0, (fixm & 1) ? 0.0 : k0, (fixm & 2) ? 0.0 : k1);
                                else { sl[L::KFF0] = (fixm & 1) ? 0.0 : k0; sl[L::KFF1] = (fixm & 2) ? 0.0 : k1; }
                                sl[O_D + 4] = 0.0;
                            }
                            w.sync();
                            w.sweep_priority(true);
                        } else {
                            if (tries_left > 0) { --tries_left; wsp = wsn; }
                            else pol = 0;        // the interior-point iteration goes on from where it was
                            w.stamp(13);
                            break;               // (no refinement of a rejected solve)
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < 8; ++m) z[m] += dzs[m] + dzl[m];
                    }
                } else if (pass == 0) {
                    // E-B: affine step lengths, centring parameter, corrector right-hand side
#pragma unroll
                    for (int m = 0; m < 8; ++m) dzs[m] = dzl[m];
                    double r_loc = 0.0;
                    double dtl[4], dtu[4], dll[4], dlu[4], itl[4], itu[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double wv = NMPC_ZQ(z, q), dw = NMPC_ZQ(dzl, q);
                        const double rtl = mq[q] ? (wv - (lb[q] - cur[q]) - tl[q]) : 0.0;
                        const double rtu = mq[q] ? ((ub[q] - cur[q]) - wv - tu[q]) : 0.0;
                        itl[q] = w.rcp(tl[q]);
                        itu[q] = w.rcp(tu[q]);
                        dtl[q] = mq[q] ? (dw + rtl) : 0.0;
                        dtu[q] = mq[q] ? (-dw + rtu) : 0.0;
                        dll[q] = -ll[q] - ll[q] * itl[q] * dtl[q];
                        dlu[q] = -lu[q] - lu[q] * itu[q] * dtu[q];
                        // largest -d/v over the positive variables (step to the boundary = 1/that)
                        r_loc = fmax(r_loc, fmax(-dtl[q] * itl[q], -dtu[q] * itu[q]));
                        r_loc = fmax(r_loc, mq[q] ? fmax(1.0 + dtl[q] * itl[q], 1.0 + dtu[q] * itu[q]) : 0.0);
                    }
                    const double a_aff = w.uniform(w.rcp(fmax(1.0, w.max(r_loc))));
                    double ma = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        ma += (tl[q] + a_aff * dtl[q]) * (ll[q] + a_aff * dll[q]) +
                              (tu[q] + a_aff * dtu[q]) * (lu[q] + a_aff * dlu[q]);
                    mu_aff = w.uniform(w.sum(ma) * inv_mcount);
                    const double sr = mu_aff * w.rcp(mu);
                    smu = w.uniform(sr * sr * sr * mu);
                    // Mehrotra's second-order term, damped when the affine step is almost fully
                    // blocked (undamped it over-corrects and the iteration can cycle)
                    csc = w.uniform(fmin(1.0, 4.0 * a_aff));
                    // (the pick-up reads above are ordered before these stores by the reductions)
                    if (act) {
                        double *sl = stg + tid * ST + O_D;
                        double dG[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double ctl = smu - csc * dtl[q] * dll[q], ctu = smu - csc * dtu[q] * dlu[q];
                            dG[q] = mq[q] ? (-ctl * itl[q] + ctu * itu[q]) : 0.0;
                        }
                        st2(sl, dG[0], dG[1]); st2(sl + 2, dG[2], dG[3]);
                        st2(sl + 4, 0.0, 0.0);     // the zero operands of the second pass (read next to dG): the
                        st2(sl + 6, 0.0, 0.0);     // corrector changes only the bounded components of the gradient
                    }
                    w.sync();
                    w.sweep_priority(true);
                } else {
                    // E-C: combined direction, step length, update
                    double r_loc = 0.0;
                    double dtl[4], dtu[4], dll[4], dlu[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double wv = NMPC_ZQ(z, q), dwa = NMPC_ZQ(dzs, q);
                        const double rtl = mq[q] ? (wv - (lb[q] - cur[q]) - tl[q]) : 0.0;
                        const double rtu = mq[q] ? ((ub[q] - cur[q]) - wv - tu[q]) : 0.0;
                        const double itl = w.rcp(tl[q]), itu = w.rcp(tu[q]);
                        // corrector targets from the affine step (recomputed, not kept in registers)
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
                    // active-set classification for the polish: where the full step (not cut back to stay interior)
                    // would leave slack below multiplier -- the limit each pair is heading for
                    cls = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (mq[q] && tl[q] + dtl[q] < ll[q] + dll[q]) cls |= 1 << (2 * q);
                        else if (mq[q] && tu[q] + dtu[q] < lu[q] + dlu[q]) cls |= 2 << (2 * q);
                    }
                    const double a_max = w.rcp(w.max(r_loc));
                    const double tau = fmax(pb.tau, 1.0 - mu_aff);
                    const double alpha = w.uniform(fmin(1.0, tau * a_max));
#pragma unroll
                    for (int m = 0; m < 8; ++m) z[m] += alpha * (dzs[m] + dzl[m]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        tl[q] += alpha * dtl[q]; tu[q] += alpha * dtu[q];
                        ll[q] += alpha * dll[q]; lu[q] += alpha * dlu[q];
                    }
#pragma unroll
                    for (int m = 0; m < 6; ++m) rb[m] *= (1.0 - alpha);
                    lin_scale = w.uniform(lin_scale * (1.0 - alpha));
                    ++it;
                    w.tick(17);
                }
                w.stamp(pol ? (pass == 0 ? 11 : pass == 1 ? 13 : 15) : (pass == 0 ? 4 : 7));
            }
            if (polished) { qp_status = 0; break; }
        }
        // what the next QP of this instance may try first (see above)
        if (polished) {
            int mm = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (the iterate's own set again: cheaper than a register held through the whole iteration)
                const int s0 = !mq[q] ? 0 : lb[q] - cur[q] >= -pb.warm_eps ? 1 : ub[q] - cur[q] <= pb.warm_eps ? 2 : 0;
                mm += NMPC_WSQ(wsp, q) != s0;
            }
            hint = w.uniform(1.0 + w.sum((double)mm));
        } else {
            hint = 0.0;
        }
        it += n_pol;    // reported count: factorisations (interior-point iterations + active-set solves)
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
        if (tid == 0) {
#pragma unroll
            for (int m = 0; m < 6; ++m) xb[m] = io.x0[m];
        }
        status = qp_status;
        if (qp_status != 0 || !W::kSqp) { ++sqp_it; break; }
        // ---- SQP mode: multipliers of this QP for the next KKT test.  Costates by the adjoint
        // recursion of the QP's x-stationarity: pi_N = 0, pi_k = H_xx dx_k + q_k + A_k' pi_{k+1}
        // (+ bound multipliers on delta, thr), k = N-1 .. 1 -- one more serial sweep, with the
        // stage data of THIS linearisation still in LDS
        if (act) {
            double *sl = stg + tid * ST + O_D;
            st2(sl - O_D + L::LAM0, ll[0], ll[1]); st2(sl - O_D + L::LAM1, ll[2], ll[3]);       // bound multipliers
            st2(sl - O_D + L::LAM2, lu[0], lu[1]); st2(sl - O_D + L::LAM3, lu[2], lu[3]);
            st2(sl + 0, Q00 * z[2] + Q01 * z[3] + Q02 * z[4] + g[2], Q01 * z[2] + Q11 * z[3] + g[3]);
            st2(sl + 2, Q02 * z[2] + Q22 * z[4] + g[4], Q33 * z[5] + g[5]);
            st2(sl + 4, Q44 * z[6] + g[6] - ll[2] + lu[2], Q55 * z[7] + g[7] - ll[3] + lu[3]);
        }
        w.sync();
        if (w.sweep_wave()) {
            // lanes 0..5 own component c of the costate (opaque: what depends on the lane alone is set up
            // here, per sweep, and not kept in registers through the whole SQP loop)
            const int c = w.opaque(lane < 6 ? lane : 5);
            const double a3 = c == 3 ? a33 : c == 5 ? a35 : 0.0;           // A[v][c]
            const double a4 = c == 4 ? 1.0 : 0.0, a5 = c == 5 ? 1.0 : 0.0;  // A[delta][c], A[thr][c]
            double pi = 0.0;
#pragma unroll 1
            for (int k = N - 1; k >= 1; --k) {
                double *sk = stg + k * ST;
                const double *col = sk + GS * c + G_XY;     // rows X,Y,psi of column c (c >= 2)
                const double a0 = c == 0 ? 1.0 : c == 1 ? 0.0 : col[0];
                const double a1 = c == 0 ? 0.0 : c == 1 ? 1.0 : col[1];
                const double a2 = c < 2 ? 0.0 : col[2];
                const double p0 = w.readlane(pi, 0), p1 = w.readlane(pi, 1), p2 = w.readlane(pi, 2),
                             p3 = w.readlane(pi, 3), p4 = w.readlane(pi, 4), p5 = w.readlane(pi, 5);
                pi = sk[O_D + c] + a0 * p0 + a1 * p1 + a2 * p2 + a3 * p3 + a4 * p4 + a5 * p5;
                if (lane < 6) sk[O_D + c] = pi;
            }
        }
        w.sync();     // costates in place for the KKT test of the next iterate
    }

    // ---------------------------------------------------------------- write back
    if (has_node) {
        double *t = io.traj + tid * 8;
#pragma unroll
        for (int m = 0; m < 6; ++m) t[m] = xb[m];
        t[6] = act ? ub_[0] : (status == 0 ? hint : 0.0);     // node N: the warm-start hint of the polish
        t[7] = act ? ub_[1] : 0.0;
    }
    const double u00 = w.readlane(ub_[0], 0), u01 = w.readlane(ub_[1], 0);
    const double d1 = w.readlane(xb[4], 1), t1 = w.readlane(xb[5], 1);
    if (tid == 0) {
        io.ctrl[0] = u00; io.ctrl[1] = u01; io.ctrl[2] = d1; io.ctrl[3] = t1;
        io.ctrl[4] = cost_out;
        io.ctrl[5] = (double)(status + 16 * total_ipm);
        *io.status = status;
        *io.iters = total_ipm;
        *io.sqp_iters = sqp_it;
        *io.cost = cost_out;
    }
    w.stamp(16);
    w.tick(19);
    w.stamp_flush();
}

#undef NMPC_ZQ
#undef NMPC_VCOL
#undef NMPC_KOFF
#undef NMPC_WSQ

}  // namespace nmpcw
