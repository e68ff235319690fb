// mpc::util::getTestTrack over nothing but <cmath>: mirrors include/mpc_local_planner/utilities.h:100 /
// src/utilities.cpp:59-72, the demo track the reference's ControlSys is constructed with
// (include/mpc_local_planner/ControlSys.h:29-30).  The rest of that file is ROS glue (message converters,
// parameter getters, the logger) and out of scope.
#pragma once

#include <cmath>
#include <vector>

#include "types.hpp"

namespace mpc {
namespace util {

// A closed figure traced by an angle that ACCUMULATES 2*pi/400 in double precision until it reaches 2*pi:
// round-off makes that 401 points, the last one a near-duplicate of the first (1e-14 apart) -- the count
// and the seam are part of what the path front-end sees, so the accumulation is kept as it is.
inline std::vector<Point> getTestTrack() {
    constexpr double kTwoPi = 2.0 * 3.14159265358979323846;
    const double step = kTwoPi / 400.0, radius = 20.0;
    std::vector<Point> track;
    track.reserve(401);
    double theta = 0.0;
    while (theta < kTwoPi) {
        track.emplace_back(2.0 * radius * std::cos(theta),
                           radius * std::sin(theta) + radius / 1.2 * std::sin(3.0 * theta));
        theta += step;
    }
    return track;
}

}  // namespace util
}  // namespace mpc
