// C++ mirror classes over the C ABI: reads "x0[6] p[6]" rows from argv[1], solves them one at a
// time with mpc::Acados::PathTracking::solve (three calls: cold + 2 warm, like the reference's
// static solver) and as one batch with solveBatch, and prints the trajectories for the Python
// test to compare with the oracle.  Needs a GPU.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <vector>

#include <cmath>

#include "mpc_local_planner_amd/PathTrackingSys.hpp"
#include "mpc_local_planner_amd/utilities.hpp"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int N = atoi(argv[2]);
    std::ifstream in(argv[1]);
    std::vector<mpc::State> states;
    std::vector<mpc::Acados::PathTrackingParams> params;
    double v[12];
    while (in >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5] >> v[6] >> v[7] >> v[8] >> v[9] >> v[10] >> v[11]) {
        states.emplace_back(v, 6);
        params.emplace_back(std::vector<double>(v + 6, v + 12));
    }
    const int B = (int)states.size();
    // batch
    mpc::Acados::PathTracking batch{states[0], N, 0.35, B};
    nmpc_reset(batch.handle(), nullptr, B);     // every instance cold from its own x0
    auto rb = batch.solveBatch(states, params);
    for (int b = 0; b < B; ++b) {
        printf("batch %d %d %g", b, (int)rb[b].success, rb[b].cost);
        for (auto &o : rb[b].mpcHorizon)
            printf(" %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", o.x.x, o.x.y, o.x.psi, o.x.vel, o.x.delta,
                   o.x.throttle, o.u.deltaDot, o.u.throttleDot);
        printf("\n");
    }
    // single instance, reference protocol: constructed from the first state, then solve() thrice
    mpc::Acados::PathTracking one{states[0], N};
    for (int call = 0; call < 3; ++call) {
        auto r = one.solve(states[0], params[0]);
        printf("single %d %d %g", call, (int)r.success, r.cost);
        for (auto &o : r.mpcHorizon)
            printf(" %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", o.x.x, o.x.y, o.x.psi, o.x.vel, o.x.delta,
                   o.x.throttle, o.u.deltaDot, o.u.throttleDot);
        printf("\n");
        if (r.mpcHorizon.size() != (size_t)N || r.u0.deltaDot != r.mpcHorizon[0].u.deltaDot) return 3;
    }
    // PathTrackingSys on util::getTestTrack (config 1: tests/testMpc.h:13 on the reference's track): state on the track -> world-frame horizon
    {
        const std::vector<mpc::Point> track = mpc::util::getTestTrack();   // the reference's own demo track (401 points)
        for (auto &pt : track) printf("track %.17g %.17g\n", pt.x, pt.y);
        mpc::PathTrackingSys sys{track, N};
        const mpc::State car{41, 0, M_PI / 2, 5, 0, 0};          // tests/testMpc.h:13
        auto r = sys.solve(car, 0.0, 5.0);
        printf("pts %d %.17g", (int)r.success, sys.lastRotation());
        for (auto &o : r.mpcHorizon) printf(" %.17g %.17g", o.x.x, o.x.y);
        printf("\n");
        if (r.mpcHorizon.size() != (size_t)N) return 5;
        // a failed solve through solve(state, params) (NaN measurement -> status 1 -> reInit) must leave
        // the track, and everything else the handle holds, in place: the next solve(state, pitch, vRef)
        // works and starts from the cold guess again (src/AcadosSolver.cpp:56-59)
        const mpc::State broken{0, 0, 0, std::nan(""), 0, 0};
        auto rf = sys.solve(broken, params[0]);
        printf("failed %d\n", (int)rf.success);
        auto r2 = sys.solve(car, 0.0, 5.0);
        printf("pts2 %d %.17g", (int)r2.success, sys.lastRotation());
        for (auto &o : r2.mpcHorizon) printf(" %.17g %.17g", o.x.x, o.x.y);
        printf("\n");
    }
    // the three-argument PathTrackingParams constructor takes anything indexable [0..3], as
    // src/PathTrackingSys.cpp:21 passes an Eigen::Vector4d
    {
        struct Vec4Like { double v[4]; double operator[](int i) const { return v[i]; } };
        const Vec4Like c{{0.5, -0.25, 0.125, -0.0625}};
        const mpc::Acados::PathTrackingParams pp{c, 0.01, 4.5};
        const auto v = pp.toVec();
        if (v.size() != 6 || v[0] != 0.5 || v[1] != -0.25 || v[2] != 0.125 || v[3] != -0.0625 || v[4] != 0.01 ||
            v[5] != 4.5)
            return 6;
        const double raw[4] = {1, 2, 3, 4};
        const mpc::Acados::PathTrackingParams pq{raw, 0.0, 5.0};
        if (pq.polyCoeffs[3] != 4.0) return 6;
    }
    // a parameter file with nlp_solver_type "SQP" runs full SQP (acados' defaults: 100 iterations, 1e-6),
    // as the key selects the generated solver in the reference (acados/mpcPathTracking.py:126)
    if (argc > 3) {
        mpc::Acados::PathTracking sq{states[0], std::string(argv[3])};
        auto r = sq.solve(states[0], params[0]);
        printf("sqpfile %d %d %d", (int)r.success, sq.horizon(), sq.runParams().mode);
        for (auto &o : r.mpcHorizon)
            printf(" %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", o.x.x, o.x.y, o.x.psi, o.x.vel, o.x.delta,
                   o.x.throttle, o.u.deltaDot, o.u.throttleDot);
        printf("\n");
    }
    // reInit after a failure with a FINITE state (src/AcadosSolver.cpp:56-59, 8-13): the guess is built from
    // the state of the failing call, and the next solve starts from it whatever it measures.  An
    // interior-point limit of 3 iterations makes every solve fail with status 4 (the step is still taken).
    if (B >= 2) {
        nmpc_config c3;
        nmpc_default_config(&c3, N, 0.35);
        c3.ipm_max_iter = 3;
        mpc::Acados::PathTracking lim{states[0], c3};
        auto r1 = lim.solve(states[0], params[0]);
        auto r2 = lim.solve(states[1], params[0]);
        printf("reinit %d %d", (int)r1.success, (int)r2.success);
        for (auto &o : r2.mpcHorizon)
            printf(" %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", o.x.x, o.x.y, o.x.psi, o.x.vel, o.x.delta,
                   o.x.throttle, o.u.deltaDot, o.u.throttleDot);
        printf("\n");
    }
    // solveBatch at the headline size (4096 vehicles): host time of one call, MPCReturn packing included
    // (the trajectories come through nmpc_get_view: split on the device, read in place from pinned memory)
    {
        const int BB = 4096;
        std::vector<mpc::State> st;
        std::vector<mpc::Acados::PathTrackingParams> pr;
        for (int b = 0; b < BB; ++b) { st.push_back(states[b % B]); pr.push_back(params[b % B]); }
        mpc::Acados::PathTracking big{states[0], N, 0.35, BB};
        double best = 1e30;
        size_t n_ok = 0;
        for (int call = 0; call < 4; ++call) {
            nmpc_reset(big.handle(), nullptr, BB);
            const auto t0 = std::chrono::steady_clock::now();
            auto r = big.solveBatch(st, pr);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (call > 0 && ms < best) best = ms;
            n_ok = 0;
            for (auto &e : r) n_ok += e.success && e.mpcHorizon.size() == (size_t)N;
        }
        printf("batchtime %d %zu %.4f\n", BB, n_ok, best);
    }
    // error behaviour: an impossible horizon throws from the constructor (AcadosPathTracking.cpp:42-47)
    try {
        mpc::Acados::PathTracking bad{states[0], 1000};
        return 4;
    } catch (const std::runtime_error &) {
        printf("ctor-throws ok\n");
    }
    return 0;
}
