// mpc::PathTrackingSys over the C ABI: mirrors include/mpc_local_planner/PathTrackingSys.h:15-86 and
// src/PathTrackingSys.cpp (same public methods and return packing), with the track window search,
// the four rotated cubic fits, the frame transform, the SQP-RTI solve and the un-rotation of the
// predicted positions all running on the GPU.  ROS publishing of the fitted polynomial
// (src/PathTrackingSys.cpp:43-54) is out of scope; the coefficients are available through
// lastCoeffs()/lastRotation() for whoever wants to publish them.
#pragma once

#include <vector>

#include "AcadosSolver.hpp"

namespace mpc {

class PathTrackingSys {
   public:
    // N, dt: the reference bakes them in at codegen (config/mpc_params.yaml:35,41)
    explicit PathTrackingSys(const std::vector<Point> &track, int N = 30, double dt = 0.35, int capacity = 1,
                             int device = 0)
        : solver_{State{}, N, dt, capacity, device}, capacity_{capacity} {
        nmpc_reset(solver_.handle(), nullptr, capacity);     // the static solver starts from the first state seen
        setTrack(track);
    }

    void setTrack(const std::vector<Point> &newTrack) {
        track_ = newTrack;
        std::vector<double> xy(track_.size() * 2);
        for (size_t i = 0; i < track_.size(); ++i) { xy[2 * i] = track_[i].x; xy[2 * i + 1] = track_[i].y; }
        check(nmpc_set_track(solver_.handle(), xy.data(), static_cast<int>(track_.size())), "nmpc_set_track");
    }

    std::vector<Point> getTrack() const { return track_; }

    // src/PathTrackingSys.cpp:17-56
    MPCReturn solve(const State &state, double pitch, double vRef) {
        return solveBatch({state}, {pitch}, {vRef}).at(0);
    }

    // src/PathTrackingSys.cpp:58-61 (solver-frame state and ready-made parameters)
    MPCReturn solve(const State &state, const Acados::PathTrackingParams &params) {
        return solver_.solve(state, params);
    }

    // B = states.size() <= capacity vehicles on the same track in one launch each of
    // front-end, solve and un-rotation
    std::vector<MPCReturn> solveBatch(const std::vector<State> &states, const std::vector<double> &pitch,
                                      const std::vector<double> &vRef) {
        const int B = static_cast<int>(states.size()), N = solver_.horizon();
        if (B < 1 || B > capacity_ || pitch.size() != states.size() || vRef.size() != states.size())
            throw std::invalid_argument("solveBatch: batch size must be 1..capacity and match pitch/vRef");
        const auto start = std::chrono::high_resolution_clock::now();
        std::vector<double> ws(static_cast<size_t>(B) * 6), pv(static_cast<size_t>(B) * 2);
        for (int b = 0; b < B; ++b) {
            const auto s = states[static_cast<size_t>(b)].toArray();
            for (int i = 0; i < 6; ++i) ws[static_cast<size_t>(b) * 6 + i] = s[static_cast<size_t>(i)];
            pv[static_cast<size_t>(b) * 2] = pitch[static_cast<size_t>(b)];
            pv[static_cast<size_t>(b) * 2 + 1] = vRef[static_cast<size_t>(b)];
        }
        nmpc_handle *h = solver_.handle();
        check(nmpc_frontend(h, ws.data(), pv.data(), B), "nmpc_frontend");
        check(nmpc_solve(h, B, NMPC_MODE_RTI, 1, 0.0), "nmpc_solve");
        std::vector<double> x(static_cast<size_t>(B) * (N + 1) * 6), u(static_cast<size_t>(B) * N * 2),
            xy(static_cast<size_t>(B) * (N + 1) * 2);
        std::vector<int32_t> status(static_cast<size_t>(B));
        check(nmpc_get(h, B, x.data(), u.data(), status.data(), nullptr, nullptr, nullptr), "nmpc_get");
        check(nmpc_get_world_path(h, B, xy.data()), "nmpc_get_world_path");
        coeffs_.assign(static_cast<size_t>(B) * 6, 0.0);
        rotation_.assign(static_cast<size_t>(B), 0.0);
        check(nmpc_get_frontend(h, B, nullptr, coeffs_.data(), rotation_.data()), "nmpc_get_frontend");
        const auto end = std::chrono::high_resolution_clock::now();
        const double ms = static_cast<double>(
            std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count());
        std::vector<MPCReturn> out(static_cast<size_t>(B));
        for (int b = 0; b < B; ++b) {
            MPCReturn &r = out[static_cast<size_t>(b)];
            r.mpcHorizon.resize(static_cast<size_t>(N));
            for (int k = 0; k < N; ++k) {
                const double *xs = &x[(static_cast<size_t>(b) * (N + 1) + k) * 6];
                State st{xs, 6};
                st.x = xy[(static_cast<size_t>(b) * (N + 1) + k) * 2];        // rotated back and shifted
                st.y = xy[(static_cast<size_t>(b) * (N + 1) + k) * 2 + 1];    // (psi stays in the solver frame)
                r.mpcHorizon[static_cast<size_t>(k)] =
                    OptVariables{st, Input{&u[(static_cast<size_t>(b) * N + k) * 2], 2}};
            }
            r.u0 = r.mpcHorizon.at(0).u;
            r.cost = -1;
            r.success = status[static_cast<size_t>(b)] == 0;
            r.computeTime = ms;
        }
        return out;
    }

    // polynomial of instance b in its rotated car frame (what the reference publishes as
    // "interpolated_path") and that frame's rotation
    std::array<double, 4> lastCoeffs(int b = 0) const {
        return {coeffs_.at(static_cast<size_t>(b) * 6), coeffs_.at(static_cast<size_t>(b) * 6 + 1),
                coeffs_.at(static_cast<size_t>(b) * 6 + 2), coeffs_.at(static_cast<size_t>(b) * 6 + 3)};
    }
    double lastRotation(int b = 0) const { return rotation_.at(static_cast<size_t>(b)); }

   private:
    static void check(int rc, const char *what) {
        if (rc != NMPC_OK) throw std::runtime_error{std::string(what) + " failed: " + nmpc_last_error()};
    }
    Acados::PathTracking solver_;
    int capacity_;
    std::vector<Point> track_;
    std::vector<double> coeffs_, rotation_;
};

}  // namespace mpc
