// C++ mirror classes over the C ABI: reads "x0[6] p[6]" rows from argv[1], solves them one at a
// time with mpc::Acados::PathTracking::solve (three calls: cold + 2 warm, like the reference's
// static solver) and as one batch with solveBatch, and prints the trajectories for the Python
// test to compare with the oracle.  Needs a GPU.
#include <cstdio>
#include <fstream>
#include <vector>

#include <cmath>

#include "mpc_local_planner_amd/PathTrackingSys.hpp"

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
    // PathTrackingSys on an ellipse-like track: state on the track -> world-frame horizon
    {
        std::vector<mpc::Point> track;
        for (int i = 0; i < 400; ++i) {
            const double th = 2 * M_PI * i / 400.0;
            track.emplace_back(40 * cos(th), 20 * sin(th) + 20 / 1.2 * sin(3 * th));
        }
        mpc::PathTrackingSys sys{track, N};
        const mpc::State car{41, 0, M_PI / 2, 5, 0, 0};          // tests/testMpc.h:13
        auto r = sys.solve(car, 0.0, 5.0);
        printf("pts %d %.17g", (int)r.success, sys.lastRotation());
        for (auto &o : r.mpcHorizon) printf(" %.17g %.17g", o.x.x, o.x.y);
        printf("\n");
        if (r.mpcHorizon.size() != (size_t)N) return 5;
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
