// emu_wave.cpp -- CPU thread emulator for mpc_local_planner_amd/csrc/nmpc_wave_solver.h.
// TEST INFRASTRUCTURE ONLY (built by tests/emu/Makefile, loaded by tests/emu_binding.py).
// It runs the *same source* the gfx950 kernel runs, with the 64 (N <= 63) or 128 threads of an
// instance's workgroup as ucontext coroutines.  Every wave policy call that exchanges data is a
// real barrier here (wave-wide or workgroup-wide, as on the GPU), so a missing or misplaced
// exchange point shows up as a wrong result or a deadlock report instead of passing by luck of
// the lock-step hardware.  The product never links this file.
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define NMPC_DEV inline
#include "../../mpc_local_planner_amd/csrc/nmpc_wave_solver.h"
#include "../../mpc_local_planner_amd/csrc/nmpc_quad_solver.h"

namespace {

constexpr int WAVE = 64;
constexpr int MAXT = 320;
constexpr size_t STACK = 512 * 1024;

struct Barrier {
    int count = 0;
    long gen = 0;
};

struct Sched {
    ucontext_t main_ctx;
    ucontext_t ctx[MAXT];
    std::vector<char> stacks;
    int nt = WAVE;              // threads of the instance
    bool done[MAXT];
    double *lds = nullptr;
    double tmp[MAXT];
    Barrier wg, wave[MAXT / WAVE];
    long spins = 0;
    bool deadlock = false;
    bool sqp = false;           // SQP mode (the host picks the instantiation by mode, as nmpc_solve does)
    const nmpcw::Problem *pb = nullptr;
    const nmpcw::InstanceIO *io = nullptr;
    // four instances per workgroup (nmpc_quad_solver.h)
    const nmpcq::Batch *batch = nullptr;
    int itmp[MAXT / WAVE];
    int sweeper = 0;
};

Sched *g_s = nullptr;

void yield_from(int t)
{
    // round robin over the threads that have not finished
    Sched &s = *g_s;
    for (int i = 1; i <= s.nt; ++i) {
        const int n = (t + i) % s.nt;
        if (n == t) return;     // nobody else can run
        if (!s.done[n]) {
            swapcontext(&s.ctx[t], &s.ctx[n]);
            return;
        }
    }
}

void barrier(Barrier &b, int nthreads, int t)
{
    const long gen = b.gen;
    if (++b.count == nthreads) {
        b.count = 0;
        ++b.gen;
        g_s->spins = 0;
    }
    // always give the others a turn: exchange points double as scheduling points
    do {
        yield_from(t);
        if (b.gen == gen && ++g_s->spins > 50000000L) {
            g_s->deadlock = true;
            setcontext(&g_s->main_ctx);
        }
    } while (b.gen == gen);
}

template <bool SQP, bool TWO = false, bool COMPACT = false>
struct EmuWave {
    using Layout = typename std::conditional<COMPACT, nmpcw::Layout48, nmpcw::Layout54>::type;
    static constexpr bool kSqp = SQP;   // same six instantiations as the GPU policy
    static constexpr bool kNeighbourLds = TWO;
    int tid_;
    int tid() const { return tid_; }
    int lane() const { return tid_ & (WAVE - 1); }
    bool sweep_wave() const { return tid_ < WAVE; }
    bool uniform_flag(bool b) { return b; }
    double *lds() { return g_s->lds; }
    int opaque(int v) { return v; }
    double rounded(double v) { return v; }
    void stamp(int) {}              // (phase timeline: debug builds of the GPU policy only)
    void tick(int) {}
    void stamp_init(double *) {}
    void stamp_flush() {}
    void sched_fence() {}
    void fwd_fetch(const double *p1, const double *p3, const double *p2, double (&d)[9])
    {
        for (int c = 0; c < 6; ++c) d[c] = p1[nmpcw::GS * c];
        d[6] = p3[0];
        d[7] = p3[2];
        d[8] = *p2;
    }
    // one forward-sweep stage on the data in `cur` while `nxt` is fetched from the next stage
    void fwd_stage(double (&cur)[9], double (&nxt)[9], const double *p1, const double *p3, const double *p2,
                   double *sw, double &val)
    {
        fwd_fetch(p1, p3, p2, nxt);
        const double start = cur[8];
        cur[8] = -1e300 * 1e10;     // the GPU policy accumulates into it: consumed
        double vA = fmac_row<0>(start, val, cur[6]);
        double vB = fmac_row<1>(0.0, val, cur[7]);
        vA = fmac_row<2>(vA, val, cur[2]);
        vB = fmac_row<3>(vB, val, cur[3]);
        vA = fmac_row<4>(vA, val, cur[4]);
        vB = fmac_row<5>(vB, val, cur[5]);
        val = vA + vB;
        val = fmac_row<6>(val, val, cur[0]);
        val = fmac_row<7>(val, val, cur[1]);
        *sw = val;
    }
    void sweep_drain() {}
    void sweep_priority(bool) {}
    double uniform(double v) { return v; }
    void bwd_fetch(const double *tv, const double *dg, const double *ab, double (&d)[6])
    {
        d[0] = tv[0]; d[1] = tv[1]; d[2] = tv[2];
        d[3] = *dg;
        d[4] = ab[0]; d[5] = ab[1];
    }
    void bwd_stage(double (&cur)[6], double (&nxt)[6], const double *tv, const double *dg, const double *ab,
                   double *kfw, double cb0, double cb1, double cb2, double keep, double &res)
    {
        bwd_fetch(tv, dg, ab, nxt);
        const double g0 = cur[3];
        cur[3] = -1e300 * 1e10;     // consumed, as on the GPU
        double hA = fmac_row<2>(g0, res, cur[0]);
        double hB = fmac_row<3>(0.0, res, cur[1]);
        hA = fmac_row<4>(hA, res, cur[2]);
        hB = fmac_row<5>(hB, res, cb0);
        hA = fmac_row<6>(hA, res, cb1);
        hB = fmac_row<7>(hB, res, cb2);
        const double hc = hA + hB;
        res = keep * hc;
        res = fmac_row<0>(res, hc, cur[4]);
        res = fmac_row<1>(res, hc, cur[5]);
        *kfw = res;
    }
    void bwd_fetch2(const double *tv, const double *dg, const double *ab, const double *ab2, double (&d)[6])
    {
        d[0] = tv[0]; d[1] = tv[1]; d[2] = tv[2];
        d[3] = *dg;
        d[4] = ab[0]; d[5] = ab2[0];
    }
    void bwd_stage2(double (&cur)[6], double (&nxt)[6], const double *tv, const double *dg, const double *ab,
                    const double *ab2, double *kfw, double cb0, double cb1, double cb2, double keep, double &res)
    {
        double nx[6];
        bwd_fetch2(tv, dg, ab, ab2, nx);
        const double g0 = cur[3];
        cur[3] = -1e300 * 1e10;     // consumed, as on the GPU
        double hA = fmac_row<2>(g0, res, cur[0]);
        double hB = fmac_row<3>(0.0, res, cur[1]);
        hA = fmac_row<4>(hA, res, cur[2]);
        hB = fmac_row<5>(hB, res, cb0);
        hA = fmac_row<6>(hA, res, cb1);
        hB = fmac_row<7>(hB, res, cb2);
        const double hc = hA + hB;
        res = keep * hc;
        res = fmac_row<0>(res, hc, cur[4]);
        res = fmac_row<1>(res, hc, cur[5]);
        *kfw = res;
        for (int i = 0; i < 6; ++i) nxt[i] = nx[i];
    }
    template <int C>
    double fmac_row(double acc, double src, double coef)   // lane C of the caller's 16-lane row
    {
        g_s->tmp[tid_] = src;
        wsync();
        const double r = fma(g_s->tmp[(tid_ & ~15) + C], coef, acc);
        wsync();
        return r;
    }
    void sync() { barrier(g_s->wg, g_s->nt, tid_); }
    void wsync() { barrier(g_s->wave[tid_ / WAVE], WAVE, tid_); }
    void sync_lockstep() { wsync(); }
    double rcp(double x) { return 1.0 / x; }
    double readlane(double v, int src)        // within the calling wave
    {
        g_s->tmp[tid_] = v;
        wsync();
        const double r = g_s->tmp[(tid_ / WAVE) * WAVE + src];
        wsync();
        return r;
    }
    double shfl_down1(double v)                // across the whole instance
    {
        g_s->tmp[tid_] = v;
        sync();
        const double r = g_s->tmp[tid_ + 1 < g_s->nt ? tid_ + 1 : tid_];
        sync();
        return r;
    }
    double sum(double v)
    {
        g_s->tmp[tid_] = v;
        sync();
        double t[MAXT];
        memcpy(t, g_s->tmp, sizeof(t));
        const int nt = g_s->nt;
        // a butterfly like the GPU policy (the association differs: round-off only)
        for (int o = nt / 2; o >= 1; o >>= 1)
            for (int l = 0; l < nt; ++l)
                if ((l & o) == 0) { const double a = t[l] + t[l | o]; t[l] = a; t[l | o] = a; }
        sync();
        return t[tid_];
    }
    double max(double v)
    {
        g_s->tmp[tid_] = v;
        sync();
        double r = g_s->tmp[0];
        for (int l = 1; l < g_s->nt; ++l) r = g_s->tmp[l] > r ? g_s->tmp[l] : r;
        sync();
        return r;
    }
};

void thread_main(int t)
{
    const bool two = g_s->nt > WAVE;
    const bool compact = nmpcw::compact_layout(g_s->pb->N);      // the rule nmpc_solve launches by
    if (g_s->sqp && two && compact) {
        EmuWave<true, true, true> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    } else if (two && compact) {
        EmuWave<false, true, true> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    } else if (g_s->sqp && compact) {
        EmuWave<true, false, true> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    } else if (compact) {
        EmuWave<false, false, true> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    } else if (g_s->sqp && two) {
        EmuWave<true, true> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    } else if (g_s->sqp) {
        EmuWave<true> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    } else if (two) {
        EmuWave<false, true> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    } else {
        EmuWave<false> w{t};
        nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    }
    g_s->done[t] = true;
    for (int i = 1; i < g_s->nt; ++i) {
        const int n = (t + i) % g_s->nt;
        if (!g_s->done[n]) setcontext(&g_s->ctx[n]);
    }
    setcontext(&g_s->main_ctx);
}


// Wave policy of the four-instances-per-workgroup solver: 256 threads, wave w = instance w in the
// element-wise phases, one wave sweeps all four (row = instance).  The factor stage restates the
// instruction list of csrc/nmpc_quad_asm.inc with the row-broadcast multiply-add as its primitive.
struct EmuQuad : EmuWave<false> {
    int lane() const { return tid_ & (WAVE - 1); }
    int wave() const { return tid_ / WAVE; }
    bool sweeper() const { return wave() == 0; }             // slot 0's wave also sweeps
    void sync() { barrier(g_s->wg, g_s->nt, tid_); }
    bool uniform_flag(bool b) { return b; }
    int next_ticket(int *queue)                // one draw per wave, the same value on every lane
    {
        if (lane() == 0) g_s->itmp[wave()] = (*queue)++;
        wsync();
        const int t = g_s->itmp[wave()];
        wsync();
        return t;
    }
    double shfl_down1(double v)                // within the wave (one instance)
    {
        g_s->tmp[tid_] = v;
        wsync();
        const double r = g_s->tmp[lane() + 1 < WAVE ? tid_ + 1 : tid_];
        wsync();
        return r;
    }
    double sum(double v)
    {
        g_s->tmp[tid_] = v;
        wsync();
        double t[WAVE];
        memcpy(t, g_s->tmp + wave() * WAVE, sizeof(t));
        for (int o = WAVE / 2; o >= 1; o >>= 1)
            for (int l = 0; l < WAVE; ++l)
                if ((l & o) == 0) { const double a = t[l] + t[l | o]; t[l] = a; t[l | o] = a; }
        wsync();
        return t[lane()];
    }
    double max(double v)
    {
        g_s->tmp[tid_] = v;
        wsync();
        const double *t = g_s->tmp + wave() * WAVE;
        double r = t[0];
        for (int l = 1; l < WAVE; ++l) r = t[l] > r ? t[l] : r;
        wsync();
        return r;
    }
    double rowb(double src, int C)             // value of lane C of the caller's 16-lane row
    {
        g_s->tmp[tid_] = src;
        wsync();
        const double r = g_s->tmp[(tid_ & ~15) + C];
        wsync();
        return r;
    }
    double fmacd(double acc, double src, int C, double coef) { return fma(rowb(src, C), coef, acc); }
    void fac_fetch(const double *tj, const double *bj, double (&ba)[6])
    {
        for (int m = 0; m < 3; ++m) { ba[m] = tj[m]; ba[3 + m] = bj[m]; }
    }
    template <int OFF>
    void fac_stage(double (&P)[8], double (&H)[8], double (&ba)[6], const double *const (&ha)[8], const double *tjn,
                   const double *bjn, double *ka, double e8)
    {
        for (int i = 0; i < 8; ++i) H[i] = ha[i][OFF];
        double M[6];
        for (int i = 0; i < 6; ++i) M[i] = P[i + 2] * e8;
        for (int m = 0; m < 6; ++m)
            for (int i = 0; i < 6; ++i) M[i] = fmacd(M[i], P[i + 2], m + 2, ba[m]);
        H[2] += M[0];
        H[3] += M[1];
        H[0] = fmacd(H[0], ba[0], 0, M[0]); H[0] = fmacd(H[0], ba[1], 0, M[1]);
        H[0] = fmacd(H[0], ba[2], 0, M[2]); H[0] = fmacd(H[0], ba[4], 0, M[4]);
        H[1] = fmacd(H[1], ba[0], 1, M[0]); H[1] = fmacd(H[1], ba[1], 1, M[1]); H[1] = fmacd(H[1], ba[2], 1, M[2]);
        H[1] = fmacd(H[1], ba[3], 1, M[3]); H[1] = fmacd(H[1], ba[5], 1, M[5]);
        H[4] = fmacd(H[4], ba[0], 4, M[0]); H[4] = fmacd(H[4], ba[1], 4, M[1]); H[4] += M[2];
        H[5] = fmacd(H[5], ba[0], 5, M[0]); H[5] = fmacd(H[5], ba[1], 5, M[1]); H[5] = fmacd(H[5], ba[2], 5, M[2]);
        H[5] = fmacd(H[5], ba[3], 5, M[3]);
        H[6] = fmacd(H[6], ba[0], 6, M[0]); H[6] = fmacd(H[6], ba[1], 6, M[1]); H[6] = fmacd(H[6], ba[2], 6, M[2]);
        H[6] += M[4];
        H[7] = fmacd(H[7], ba[0], 7, M[0]); H[7] = fmacd(H[7], ba[1], 7, M[1]); H[7] = fmacd(H[7], ba[2], 7, M[2]);
        H[7] = fmacd(H[7], ba[3], 7, M[3]); H[7] += M[5];
        for (int m = 0; m < 3; ++m) { ba[m] = tjn[m]; ba[3 + m] = bjn[m]; }
        const double t11 = rowb(H[1], 1), t01 = rowb(H[0], 1);
        double det = -t01 * t01;
        det = fmacd(det, H[0], 0, t11);
        const double y = 1.0 / det;
        const double n00 = -t11 * y, n01 = t01 * y;
        const double k0 = fma(n01, H[1], n00 * H[0]);
        double k1 = n01 * H[0];
        k1 = fmacd(k1, H[0], 0, -(y * H[1]));
        for (int i = 0; i < 6; ++i) H[i + 2] = fmacd(H[i + 2], H[0], i + 2, k0);
        for (int i = 0; i < 6; ++i) H[i + 2] = fmacd(H[i + 2], H[1], i + 2, k1);
        ka[0] = k0;
        ka[1] = k1;
    }
};

void thread_main_quad(int t)
{
    EmuQuad w;
    w.tid_ = t;
    nmpcq::solve_quad(w, *g_s->pb, *g_s->batch);
    g_s->done[t] = true;
    for (int i = 1; i < g_s->nt; ++i) {
        const int n = (t + i) % g_s->nt;
        if (!g_s->done[n]) setcontext(&g_s->ctx[n]);
    }
    setcontext(&g_s->main_ctx);
}

}  // namespace

extern "C" int emu_solve_batch(int N, double dt, double wheelbase, int ipm_max_iter, double mu0,
                               double thr0, double tau, double tol_mu, double tol_lin, double tol_comp,
                               int sqp_max_iter, double sqp_tol, int B, const double *x0,
                               const double *p, const double *W, const double *bnd, double *traj,
                               double *ctrl, int *status, int *iters, int *sqp_iters, double *cost,
                               int cold, int stall_iter, double stall_lin, int polish, double polish_mu,
                               int polish_tries, double polish_tol, int warm_tries, int warm_gate, double warm_eps)
{
    // sqp_max_iter <= 0: one SQP-RTI iteration; > 0: SQP mode with that many iterations at most
    if (N < 1 || N > 127) return 2;
    nmpcw::Problem pb;
    pb.N = N; pb.dt = dt; pb.inv_wheelbase = 1.0 / wheelbase;
    pb.ipm_max_iter = ipm_max_iter; pb.mu0 = mu0; pb.thr0 = thr0; pb.tau = tau;
    pb.tol_mu = tol_mu; pb.tol_lin = tol_lin; pb.tol_comp = tol_comp;
    pb.sqp_max_iter = sqp_max_iter > 0 ? sqp_max_iter : 1; pb.sqp_tol = sqp_tol;
    pb.stall_iter = stall_iter; pb.stall_lin = stall_lin;
    pb.polish = polish; pb.polish_mu = polish_mu; pb.polish_tries = polish_tries; pb.polish_tol = polish_tol;
    pb.warm_tries = warm_tries; pb.warm_gate = warm_gate; pb.warm_eps = warm_eps;
    Sched s;
    g_s = &s;
    s.sqp = sqp_max_iter > 0;
    s.nt = WAVE * nmpcw::waves_per_instance(N);
    s.stacks.resize(STACK * s.nt);
    std::vector<double> lds((size_t)nmpcw::lds_doubles(N) + 2);
    // 16-byte aligned base, as on the GPU (the body uses paired loads)
    s.lds = lds.data() + ((reinterpret_cast<size_t>(lds.data()) & 15) ? 1 : 0);
    s.pb = &pb;
    int rc = 0;
    for (int b = 0; b < B; ++b) {
        nmpcw::InstanceIO io;
        io.x0 = x0 + (size_t)b * 6; io.p = p + (size_t)b * 6; io.W = W + (size_t)b * 7;
        io.bnd = bnd + (size_t)b * 8; io.traj = traj + (size_t)b * (N + 1) * 8;
        io.ctrl = ctrl + (size_t)b * 6; io.status = status + b; io.iters = iters + b;
        io.sqp_iters = sqp_iters + b; io.cost = cost + b; io.cold = cold; io.N = 0;
        s.io = &io;
        for (size_t i = 0; i < (size_t)nmpcw::lds_doubles(N); ++i) s.lds[i] = -1e300 * 1e10;   // poison (-inf)
        s.wg = Barrier();
        for (auto &wb : s.wave) wb = Barrier();
        s.spins = 0;
        for (int t = 0; t < s.nt; ++t) {
            s.done[t] = false;
            getcontext(&s.ctx[t]);
            s.ctx[t].uc_stack.ss_sp = s.stacks.data() + STACK * t;
            s.ctx[t].uc_stack.ss_size = STACK;
            s.ctx[t].uc_link = nullptr;
            makecontext(&s.ctx[t], (void (*)())thread_main, 1, t);
        }
        swapcontext(&s.main_ctx, &s.ctx[0]);
        if (s.deadlock) {
            fprintf(stderr, "emu: threads never met at an exchange point (instance %d)\n", b);
            rc = 1;
            break;
        }
        for (int t = 0; t < s.nt; ++t)
            if (!s.done[t]) { fprintf(stderr, "emu: thread %d did not finish\n", t); rc = 1; }
    }
    if (const char *f = getenv("EMU_DUMP_LDS")) {      // debugging aid: the LDS image after the last instance
        FILE *fp = fopen(f, "wb");
        if (fp) { fwrite(s.lds, sizeof(double), (size_t)nmpcw::lds_doubles(N), fp); fclose(fp); }
    }
    g_s = nullptr;
    return rc;
}

// SQP-RTI with the four-instances-per-workgroup solver (N <= 63): same arguments; `sweeper` = the wave
// that runs the sweeps.  The last workgroup is padded with copies of the last instance that write nothing.
extern "C" int emu_solve_batch_quad(int N, double dt, double wheelbase, int ipm_max_iter, double mu0, double thr0,
                                    double tau, double tol_mu, double tol_lin, double tol_comp, int B,
                                    const double *x0, const double *p, const double *W, const double *bnd,
                                    double *traj, double *ctrl, int *status, int *iters, int *sqp_iters,
                                    double *cost, int cold, int sweeper, int stall_iter, double stall_lin)
{
    (void)sweeper;
    if (N < 1 || N > 63) return 2;
    nmpcw::Problem pb;
    pb.N = N; pb.dt = dt; pb.inv_wheelbase = 1.0 / wheelbase;
    pb.ipm_max_iter = ipm_max_iter; pb.mu0 = mu0; pb.thr0 = thr0; pb.tau = tau;
    pb.tol_mu = tol_mu; pb.tol_lin = tol_lin; pb.tol_comp = tol_comp;
    pb.sqp_max_iter = 1; pb.sqp_tol = 0.0;
    pb.stall_iter = stall_iter; pb.stall_lin = stall_lin;
    pb.polish = 0; pb.polish_mu = 0.0; pb.polish_tries = 0; pb.polish_tol = 0.0;     // (this solver is interior point only)
    pb.warm_tries = -1; pb.warm_gate = 0; pb.warm_eps = 0.0;
    Sched s;
    g_s = &s;
    s.nt = WAVE * nmpcq::QI;
    s.stacks.resize(STACK * s.nt);
    const size_t nl = (size_t)nmpcq::quad_lds_doubles(N);
    std::vector<double> lds(nl + 2);
    s.lds = lds.data() + ((reinterpret_cast<size_t>(lds.data()) & 15) ? 1 : 0);
    s.pb = &pb;
    int rc = 0;
    // one workgroup whose four slots draw the whole batch from the ticket queue; a second one after it
    // finds the queue empty and must end at once (what surplus workgroups of a launch do)
    std::vector<unsigned char> reset((size_t)B, (unsigned char)(cold ? 1 : 0));
    int queue = 0;
    nmpcq::Batch bt;
    bt.B = B; bt.x0 = x0; bt.p = p; bt.W = W; bt.bnd = bnd; bt.w_stride = 7; bt.bnd_stride = 8;
    bt.traj = traj; bt.ctrl = ctrl; bt.cost = cost; bt.status = status; bt.iters = iters; bt.sqp_iters = sqp_iters;
    bt.reset = reset.data(); bt.order = nullptr; bt.queue = &queue;
    s.batch = &bt;
    for (int wg = 0; wg < 2 && rc == 0; ++wg) {
        for (size_t i = 0; i < nl; ++i) s.lds[i] = -1e300 * 1e10;   // poison (-inf)
        s.wg = Barrier();
        for (auto &wb : s.wave) wb = Barrier();
        s.spins = 0;
        for (int t = 0; t < s.nt; ++t) {
            s.done[t] = false;
            getcontext(&s.ctx[t]);
            s.ctx[t].uc_stack.ss_sp = s.stacks.data() + STACK * t;
            s.ctx[t].uc_stack.ss_size = STACK;
            s.ctx[t].uc_link = nullptr;
            makecontext(&s.ctx[t], (void (*)())thread_main_quad, 1, t);
        }
        swapcontext(&s.main_ctx, &s.ctx[0]);
        if (s.deadlock) {
            fprintf(stderr, "emu(quad): threads never met at an exchange point\n");
            rc = 1;
            break;
        }
        for (int t = 0; t < s.nt; ++t)
            if (!s.done[t]) { fprintf(stderr, "emu(quad): thread %d did not finish\n", t); rc = 1; }
        if (queue < B) { fprintf(stderr, "emu(quad): workgroup ended with %d of %d tickets drawn\n", queue, B); rc = 1; }
        if (const char *f = getenv("EMU_DUMP_LDS")) {  // debugging aid: the LDS image of the first workgroup at its end
            static int dumped_for = -1;
            if (dumped_for != queue || true) {
                FILE *fp = fopen(f, "ab");
                if (fp) { fwrite(s.lds, sizeof(double), nl, fp); fclose(fp); }
            }
            dumped_for = queue;
        }
    }
    g_s = nullptr;
    return rc;
}
