// emu_wave.cpp -- CPU lane emulator for mpc_local_planner_amd/csrc/nmpc_wave_solver.h.
// TEST INFRASTRUCTURE ONLY (built by tests/emu/Makefile, loaded by tests/test_emu_*.py).
// It runs the *same source* the gfx950 kernel runs, with the 64 lanes of a wave as 64
// ucontext coroutines that switch round-robin at every wave sync / cross-lane op, so the
// lane mappings, LDS layout and exchange protocol can be checked against the oracle on a
// machine without a GPU.  The product never links this file.
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NMPC_DEV inline
#include "../../mpc_local_planner_amd/csrc/nmpc_wave_solver.h"

namespace {

constexpr int WAVE = 64;
constexpr size_t STACK = 512 * 1024;

struct Sched {
    ucontext_t main_ctx;
    ucontext_t ctx[WAVE];
    std::vector<char> stacks;
    double *lds = nullptr;
    double tmp[WAVE];
    long nsync[WAVE];
    const nmpcw::Problem *pb = nullptr;
    const nmpcw::InstanceIO *io = nullptr;
};

Sched *g_s = nullptr;

struct EmuWave {
    int lane_;
    int lane() const { return lane_; }
    double *lds() { return g_s->lds; }
    int opaque(int v) { return v; }
    void sched_fence() {}
    void lds_gather_fwd(const double *kr, const double *rw, const double *rbp, double (&kk)[7],
                        double (&rr)[6], double *rbv)
    {
        for (int c = 0; c < 7; ++c) kk[c] = kr[2 * c];
        for (int c = 0; c < 6; ++c) rr[c] = rw[4 * c];
        *rbv = *rbp;
    }
    void sync()
    {
        g_s->nsync[lane_]++;
        swapcontext(&g_s->ctx[lane_], &g_s->ctx[(lane_ + 1) % WAVE]);
    }
    void sync_lockstep() { sync(); }
    double rcp(double x) { return 1.0 / x; }
    double readlane(double v, int src)
    {
        g_s->tmp[lane_] = v;
        sync();
        const double r = g_s->tmp[src];
        sync();
        return r;
    }
    double shfl_down1(double v)
    {
        g_s->tmp[lane_] = v;
        sync();
        const double r = g_s->tmp[lane_ + 1 < WAVE ? lane_ + 1 : lane_];
        sync();
        return r;
    }
    double sum(double v)
    {
        g_s->tmp[lane_] = v;
        sync();
        double t[WAVE];
        memcpy(t, g_s->tmp, sizeof(t));
        for (int o = 32; o >= 1; o >>= 1)   // a butterfly like the GPU policy (association differs: round-off only)
            for (int l = 0; l < WAVE; ++l)
                if ((l & o) == 0) { const double a = t[l] + t[l | o]; t[l] = a; t[l | o] = a; }
        sync();
        return t[lane_];
    }
    double max(double v)
    {
        g_s->tmp[lane_] = v;
        sync();
        double r = g_s->tmp[0];
        for (int l = 1; l < WAVE; ++l) r = g_s->tmp[l] > r ? g_s->tmp[l] : r;
        sync();
        return r;
    }
};

void lane_main(int lane)
{
    EmuWave w{lane};
    nmpcw::solve_instance(w, *g_s->pb, *g_s->io);
    if (lane + 1 < WAVE)
        setcontext(&g_s->ctx[lane + 1]);
    else
        setcontext(&g_s->main_ctx);
}

}  // namespace

extern "C" int emu_solve_batch(int N, double dt, double wheelbase, int ipm_max_iter, double mu0,
                               double thr0, double tau, double tol_mu, double tol_lin, double tol_comp,
                               int sqp_max_iter, double sqp_tol, int B, const double *x0,
                               const double *p, const double *W, const double *bnd, double *traj,
                               double *ctrl, int *status, int *iters, int *sqp_iters, double *cost,
                               int cold)
{
    nmpcw::Problem pb;
    pb.N = N; pb.dt = dt; pb.inv_wheelbase = 1.0 / wheelbase;
    pb.ipm_max_iter = ipm_max_iter; pb.mu0 = mu0; pb.thr0 = thr0; pb.tau = tau;
    pb.tol_mu = tol_mu; pb.tol_lin = tol_lin; pb.tol_comp = tol_comp; pb.sqp_max_iter = sqp_max_iter; pb.sqp_tol = sqp_tol;
    Sched s;
    g_s = &s;
    s.stacks.resize(STACK * WAVE);
    std::vector<double> lds((size_t)nmpcw::lds_doubles(N));
    s.lds = lds.data();
    s.pb = &pb;
    int rc = 0;
    for (int b = 0; b < B; ++b) {
        nmpcw::InstanceIO io;
        io.x0 = x0 + (size_t)b * 6; io.p = p + (size_t)b * 6; io.W = W + (size_t)b * 7;
        io.bnd = bnd + (size_t)b * 8; io.traj = traj + (size_t)b * (N + 1) * 8;
        io.ctrl = ctrl + (size_t)b * 6; io.status = status + b; io.iters = iters + b;
        io.sqp_iters = sqp_iters + b; io.cost = cost + b; io.cold = cold;
        s.io = &io;
        for (size_t i = 0; i < lds.size(); ++i) lds[i] = -1e300 * 1e10;   // poison (-inf)
        for (int l = 0; l < WAVE; ++l) {
            s.nsync[l] = 0;
            getcontext(&s.ctx[l]);
            s.ctx[l].uc_stack.ss_sp = s.stacks.data() + STACK * l;
            s.ctx[l].uc_stack.ss_size = STACK;
            s.ctx[l].uc_link = nullptr;
            makecontext(&s.ctx[l], (void (*)())lane_main, 1, l);
        }
        swapcontext(&s.main_ctx, &s.ctx[0]);
        for (int l = 1; l < WAVE; ++l)
            if (s.nsync[l] != s.nsync[0]) { fprintf(stderr, "emu: divergent sync count lane %d\n", l); rc = 1; }
    }
    g_s = nullptr;
    return rc;
}
