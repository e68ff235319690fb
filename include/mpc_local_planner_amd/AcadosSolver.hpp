// mpc::Acados::Solver / PathTrackingParams / PathTracking over the C ABI of libnmpc_hip.so.
//
// Mirrors the reference's public solver interface
//   include/mpc_local_planner/AcadosSolver.h:16-81        class Solver
//   include/mpc_local_planner/AcadosPathTracking.h:10-76  PathTrackingParams, class PathTracking
//   src/AcadosSolver.cpp:8-92, src/AcadosPathTracking.cpp:6-64   behaviour
// (same method names, argument meaning, return packing and error behaviour) with the acados
// capsule replaced by an nmpc_handle.  One object = a batch of `capacity` independent OCP
// instances on one GPU; capacity 1 is the reference's single static solver
// (src/PathTrackingSys.cpp:59).  Batch entry point: solveBatch().
//
// Header-only; link with -lnmpc_hip (mpc_local_planner_amd/libnmpc_hip.so).  No CPU fallback.
#pragma once

#include <array>
#include <chrono>
#include <cstdint>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../nmpc.h"
#include "types.hpp"

namespace mpc {
namespace Acados {

// AcadosPathTracking.h:10-47.  polyCoeffs is indexable like the reference's Eigen::Vector4d, and the
// three-argument constructor takes anything indexable [0..3] (Eigen::Vector4d, std::array, a
// pointer ...), so src/PathTrackingSys.cpp:21 compiles unchanged with or without Eigen.
struct PathTrackingParams : public Params {
    PathTrackingParams() = default;
    template <class Vec4>
    PathTrackingParams(const Vec4 &coeffs, double pitch, double vRef)
        : polyCoeffs{{static_cast<double>(coeffs[0]), static_cast<double>(coeffs[1]),
                      static_cast<double>(coeffs[2]), static_cast<double>(coeffs[3])}},
          pitch{pitch}, vRef{vRef} {}
    explicit PathTrackingParams(const std::vector<double> &v) {
        if (v.size() != size) throw std::invalid_argument("PathTrackingParams needs 6 values");
        polyCoeffs = {v[0], v[1], v[2], v[3]};
        pitch = v[4];
        vRef = v[5];
    }
    std::vector<double> toVec() const override {
        return {polyCoeffs[0], polyCoeffs[1], polyCoeffs[2], polyCoeffs[3], pitch, vRef};
    }
    std::array<double, 4> polyCoeffs{{0, 0, 0, 0}};
    double pitch = 0;
    double vRef = 0;
    static constexpr size_t size = 6;
};

// AcadosSolver.h:16-81.  The protected virtual trio (acadosSolve/init/freeAllocated) is kept so
// that further problem classes can derive exactly as PathTracking does in the reference.
class Solver {
   public:
    Solver() = default;
    virtual ~Solver() = default;
    Solver(const Solver &) = delete;             // the reference's class double-frees when copied
    Solver &operator=(const Solver &) = delete;

    // AcadosSolver.cpp:8-13 frees the capsule, creates a new one and sets the cold guess from the state of
    // the failing call: the next solve starts from x_k = that state, u_k = 0, whatever it measures then.
    // The capsule there holds ONE solver and nothing else; the handle here also holds the track,
    // per-instance weights / bounds, redirected control records and the warm iterates of the other
    // capacity-1 instances, so the same effect -- solver state of instance 0 (the one solve() drives) back
    // to that guess -- is reached without tearing the handle down: nmpc_set_iterate installs the guess and
    // clears the cold-start flag the kernel raised for the failed instance.  A state that is not finite
    // (the failure was a NaN measurement) cannot serve as a guess: then the instance is flagged cold
    // instead and starts from its NEXT measurement.
    void reInit(const State &state) {
        if (!handle_) init();
        const auto s = state.toArray();
        bool finite = true;
        for (double v : s) finite = finite && (v - v == 0.0);
        if (finite) {
            setInitGuess(state);
        } else {
            const uint8_t first = 1;
            check(nmpc_reset(handle_, &first, 1), "nmpc_reset");
        }
    }

    void setInitCondition(const State &state) {  // AcadosSolver.cpp:15-26: lbx0 = ubx0 = x0
        const auto x0 = state.toArray();
        check(nmpc_set_x0(handle_, x0.data(), 1), "nmpc_set_x0");
    }

    virtual void setParams(const Params &params) = 0;

    void setInitGuess(const State &state) {      // AcadosSolver.cpp:78-92: x_k = state, u_k = 0
        const int N = horizon();
        std::vector<double> x(static_cast<size_t>(N + 1) * 6), u(static_cast<size_t>(N) * 2, 0.0);
        const auto s = state.toArray();
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < 6; ++i) x[static_cast<size_t>(k) * 6 + i] = s[i];
        check(nmpc_set_iterate(handle_, x.data(), u.data(), 1), "nmpc_set_iterate");
    }

    MPCReturn solve(const State &state, const Params &params) {   // AcadosSolver.cpp:28-76
        const auto start = std::chrono::high_resolution_clock::now();
        const int N = horizon();
        setInitCondition(state);
        setParams(params);
        const int rc = acadosSolve();
        std::vector<double> xtraj(static_cast<size_t>(N + 1) * nx_), utraj(static_cast<size_t>(N) * nu_);
        int32_t status = 0;
        check(nmpc_get(handle_, 1, xtraj.data(), utraj.data(), &status, nullptr, nullptr, nullptr), "nmpc_get");
        if (rc != NMPC_OK) status = 4;
        if (status != 0) {
            std::cout << "ERROS: acados_solve() failed with status " << status << std::endl;
            reInit(state);
        }
        const auto end = std::chrono::high_resolution_clock::now();
        MPCReturn ret = pack(xtraj.data(), utraj.data(), N, status);
        ret.computeTime = static_cast<double>(
            std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count());
        return ret;
    }

    int horizon() const { return nmpc_horizon(handle_); }
    nmpc_handle *handle() const { return handle_; }

   protected:
    virtual int acadosSolve() = 0;
    virtual void init() = 0;
    virtual void freeAllocated() = 0;

    static void check(int rc, const char *what) {
        if (rc != NMPC_OK) {
            std::stringstream e;
            e << what << " returned status: " << rc << " (" << nmpc_last_error() << ")";
            throw std::runtime_error{e.str()};
        }
    }

    // N entries x_0..x_{N-1} with u_0..u_{N-1}; x_N is read but dropped, cost = -1
    // (AcadosSolver.cpp:64-75)
    MPCReturn pack(const double *x, const double *u, int N, int status) const {
        MPCReturn ret;
        ret.mpcHorizon.resize(static_cast<size_t>(N));
        for (int i = 0; i < N; ++i)
            ret.mpcHorizon[static_cast<size_t>(i)] =
                OptVariables{State{x + static_cast<size_t>(i) * nx_, nx_}, Input{u + static_cast<size_t>(i) * nu_, nu_}};
        ret.u0 = ret.mpcHorizon.at(0).u;
        ret.cost = -1;
        ret.success = status == 0;
        return ret;
    }

    nmpc_handle *handle_ = nullptr;              // replaces capsule_/config_/dims_/in_/out_/solver_/opts_
    size_t nx_ = 6, nu_ = 2;
};

// AcadosPathTracking.h:50-76.  N, dt, bounds and weights are run-time configuration here (the
// reference bakes config/mpc_params.yaml into generated code); defaults are the reference's.
class PathTracking : public Solver {
   public:
    explicit PathTracking(const State &state, int N = 30, double dt = 0.35, int capacity = 1, int device = 0)
        : capacity_{capacity}, device_{device} {
        check(nmpc_default_config(&cfg_, N, dt), "nmpc_default_config");
        init();
        setInitGuess(state);
    }
    PathTracking(const State &state, const nmpc_config &cfg, int capacity = 1, int device = 0)
        : cfg_{cfg}, capacity_{capacity}, device_{device} {
        init();
        setInitGuess(state);
    }
    // From a ROS parameter file shaped like config/mpc_params.yaml (what the reference reads at code
    // generation, acados/mpcPathTracking.py:83-92); throws std::runtime_error naming the bad key.
    PathTracking(const State &state, const std::string &paramsFile, int capacity = 1, int device = 0)
        : capacity_{capacity}, device_{device} {
        check(nmpc_config_from_file(paramsFile.c_str(), &cfg_, &run_), "nmpc_config_from_file");
        init();
        setInitGuess(state);
    }
    ~PathTracking() override { freeAllocated(); }
    const nmpc_run_params &runParams() const { return run_; }    // defaults unless built from a file

    void setParams(const Params &params) override {              // AcadosPathTracking.cpp:17-24
        const auto p = params.toVec();
        if (p.size() != PathTrackingParams::size) throw std::invalid_argument("expected 6 parameters");
        check(nmpc_set_params(handle_, p.data(), 1), "nmpc_set_params");
    }

    // Batch form of solve(): B = states.size() <= capacity independent instances in one launch.
    // Same protocol per instance (x0 embedding, parameters on all stages, one SQP-RTI iteration,
    // warm start without shift, cold re-initialisation of failed instances).
    std::vector<MPCReturn> solveBatch(const std::vector<State> &states,
                                      const std::vector<PathTrackingParams> &params) {
        const auto start = std::chrono::high_resolution_clock::now();
        const int B = static_cast<int>(states.size()), N = horizon();
        if (B < 1 || B > capacity_ || params.size() != states.size())
            throw std::invalid_argument("solveBatch: batch size must be 1..capacity and match params");
        std::vector<double> x0(static_cast<size_t>(B) * 6), p(static_cast<size_t>(B) * 6);
        for (int b = 0; b < B; ++b) {
            const auto s = states[static_cast<size_t>(b)].toArray();
            const PathTrackingParams &q = params[static_cast<size_t>(b)];     // (toVec() would allocate per vehicle)
            double *pr = p.data() + static_cast<size_t>(b) * 6;
            for (int i = 0; i < 6; ++i) x0[static_cast<size_t>(b) * 6 + i] = s[static_cast<size_t>(i)];
            for (int i = 0; i < 4; ++i) pr[i] = q.polyCoeffs[static_cast<size_t>(i)];
            pr[4] = q.pitch;
            pr[5] = q.vRef;
        }
        check(nmpc_set_x0(handle_, x0.data(), B), "nmpc_set_x0");
        check(nmpc_set_params(handle_, p.data(), B), "nmpc_set_params");
        check(nmpc_solve(handle_, B, run_.mode, sqpMaxIter_, sqpTol_), "nmpc_solve");
        // the horizons are read once, to pack the MPCReturns: straight from the handle's pinned host buffers
        const double *x = nullptr, *u = nullptr;
        std::vector<int32_t> status(static_cast<size_t>(B));
        check(nmpc_get(handle_, B, nullptr, nullptr, status.data(), nullptr, nullptr, nullptr), "nmpc_get");
        check(nmpc_get_view(handle_, B, &x, &u), "nmpc_get_view");
        const auto end = std::chrono::high_resolution_clock::now();
        const double ms = static_cast<double>(
            std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count());
        std::vector<MPCReturn> out(static_cast<size_t>(B));
        for (int b = 0; b < B; ++b) {
            out[static_cast<size_t>(b)] = pack(x + static_cast<size_t>(b) * (N + 1) * 6,
                                               u + static_cast<size_t>(b) * N * 2, N,
                                               status[static_cast<size_t>(b)]);
            out[static_cast<size_t>(b)].computeTime = ms;
        }
        return out;   // failed instances restart cold on their next solve (device-side reset flag)
    }

    const nmpc_config &config() const { return cfg_; }
    int capacity() const { return capacity_; }

    // Horizon sweep inside one batch: instance b solves with horizon N[b] (1..config().N); an empty
    // vector restores the handle's horizon for all.  MPCReturn::mpcHorizon of solveBatch keeps
    // config().N entries, of which the first N[b] are the instance's.
    void setHorizons(const std::vector<int32_t> &N) {
        check(nmpc_set_horizons(handle_, N.empty() ? nullptr : N.data(), static_cast<int>(N.size())), "nmpc_set_horizons");
    }

    // SQP mode only (nlp_solver_type "SQP"): iteration limit and KKT tolerance; the defaults are acados'
    // (nlp_solver_max_iter 100, nlp_solver_tol_* 1e-6), which the reference leaves untouched
    void setSqpOptions(int maxIter, double tol) { sqpMaxIter_ = maxIter; sqpTol_ = tol; }
    void setMode(int mode) { run_.mode = mode; }

   private:
    int acadosSolve() override {                                 // AcadosPathTracking.cpp:26
        // nlp_solver_type of the parameter file selects the solver, as it selects the generated one in
        // the reference (acados/mpcPathTracking.py:126)
        const int rc = nmpc_solve(handle_, 1, run_.mode, sqpMaxIter_, sqpTol_);
        return rc != NMPC_OK ? rc : nmpc_sync(handle_);
    }

    void init() override {                                       // AcadosPathTracking.cpp:28-49
        nx_ = 6;
        nu_ = 2;
        const int status = nmpc_create(&cfg_, capacity_, device_, &handle_);
        if (status) {
            std::stringstream error;
            error << "nmpc_create() returned status: " << status << " (" << nmpc_last_error() << "). Exiting.";
            std::cout << "Error: " << error.str() << std::endl;
            throw std::runtime_error{error.str()};
        }
    }

    void freeAllocated() override {                              // AcadosPathTracking.cpp:51-64
        if (!handle_) return;
        const int status = nmpc_destroy(handle_);
        handle_ = nullptr;
        if (status) std::cout << "Error: nmpc_destroy() returned status " << status << std::endl;
    }

    nmpc_config cfg_{};
    nmpc_run_params run_{NMPC_MODE_RTI, 30.0, 5.0, 1.0};
    int capacity_ = 1;
    int device_ = 0;
    int sqpMaxIter_ = 100;
    double sqpTol_ = 1e-6;
};

}  // namespace Acados
}  // namespace mpc
