// Value types of the solver boundary, mirroring the reference's
// include/mpc_local_planner/types.h:10-170 (same names, field order and constructors) so that
// code written against mpc::State / Input / OptVariables / MPCReturn / Params compiles unchanged.
// Differences: no Eigen dependency (the reference includes Eigen/Core here but uses it only for
// Vector4d in AcadosPathTracking.h), and the size-checked constructors throw instead of assert.
#pragma once

#include <array>
#include <cstddef>
#include <ostream>
#include <stdexcept>
#include <vector>

namespace mpc {

struct State {                                   // types.h:10-60, order x,y,psi,vel,delta,throttle
    State() = default;
    State(double x, double y, double psi, double vel, double delta, double throttle)
        : x{x}, y{y}, psi{psi}, vel{vel}, delta{delta}, throttle{throttle} {}
    State(const std::array<double, 6> &a) : State{a[0], a[1], a[2], a[3], a[4], a[5]} {}
    State(double const *a, size_t n) : State{} {
        if (n != 6) throw std::invalid_argument("State needs 6 values");
        x = a[0]; y = a[1]; psi = a[2]; vel = a[3]; delta = a[4]; throttle = a[5];
    }
    std::array<double, 6> toArray() const { return {x, y, psi, vel, delta, throttle}; }
    double x = 0, y = 0, psi = 0, vel = 0, delta = 0, throttle = 0;
};

struct Input {                                   // types.h:63-90, order deltaDot, throttleDot
    Input() = default;
    Input(double deltaDot, double throttleDot) : deltaDot{deltaDot}, throttleDot{throttleDot} {}
    Input(const std::array<double, 2> &a) : Input{a[0], a[1]} {}
    Input(double const *a, size_t n) : Input{} {
        if (n != 2) throw std::invalid_argument("Input needs 2 values");
        deltaDot = a[0]; throttleDot = a[1];
    }
    std::array<double, 2> toArray() const { return {deltaDot, throttleDot}; }
    double deltaDot = 0, throttleDot = 0;
};

struct OptVariables {                            // types.h:93-108
    OptVariables() = default;
    OptVariables(const State &x, const Input &u) : x{x}, u{u} {}
    State x;
    Input u;
};

struct MPCReturn {                               // types.h:111-150
    MPCReturn() = default;
    MPCReturn(const Input &u0, const std::vector<OptVariables> &mpcHorizon, double computeTime,
              double cost, bool success)
        : u0{u0}, mpcHorizon{mpcHorizon}, computeTime{computeTime}, cost{cost}, success{success} {}
    MPCReturn(const std::vector<OptVariables> &mpcHorizon, double computeTime, double cost, bool success)
        : MPCReturn{mpcHorizon.at(0).u, mpcHorizon, computeTime, cost, success} {}
    Input u0;                                    // = mpcHorizon[0].u
    std::vector<OptVariables> mpcHorizon;        // N entries: x_0..x_{N-1}, u_0..u_{N-1}
    double computeTime = 0;                      // [ms]
    double cost = 0;                             // the reference always reports -1
    bool success = false;
};

struct Point {                                   // types.h:153-165
    Point() = default;
    Point(double x, double y) : x{x}, y{y} {}
    double x = 0, y = 0;
};

struct Params {                                  // types.h:168-170
    virtual ~Params() = default;
    virtual std::vector<double> toVec() const = 0;
};

}  // namespace mpc

inline std::ostream &operator<<(std::ostream &os, const mpc::State &s) {
    return os << "x: " << s.x << ", y: " << s.y << ", psi: " << s.psi << ", vel: " << s.vel
              << ", delta: " << s.delta << ", throttle: " << s.throttle;
}
inline std::ostream &operator<<(std::ostream &os, const mpc::Input &u) {
    return os << "deltaDot: " << u.deltaDot << ", throttleDot: " << u.throttleDot;
}
inline std::ostream &operator<<(std::ostream &os, const mpc::OptVariables &v) {
    return os << "x: {" << v.x << "}, u: {" << v.u << "}";
}
