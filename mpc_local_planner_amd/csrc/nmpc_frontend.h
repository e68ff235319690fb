// nmpc_frontend.h -- per-instance path front-end and plant step (one GPU thread per instance).
//
// What it replaces (reference file:line):
//   PathTrackingSys::getTrackSection   src/PathTrackingSys.cpp:107-136  nearest point + 15 m window
//   PathTrackingSys::calcCoeffs        src/PathTrackingSys.cpp:63-79    4 frame rotations, min SSE
//   PathTrackingSys::interpolate       src/PathTrackingSys.cpp:81-105   rotate/shift points, cubic fit
//   util::polyfit                      src/utilities.cpp:39-57          Householder QR least squares
//   frame transform / un-rotation      src/PathTrackingSys.cpp:23,27-41
//   actuation rule                     src/RosMpc.cpp:81-90             stage-1 delta / throttle
//
// The cubic fit is an unpivoted Householder QR of the n x 4 Vandermonde matrix, as Eigen's
// householderQr().solve() computes it, but WITHOUT storing the matrix: the state of a row after
// the first j reflections is a function of (x_i, y_i) and of the j*(9-j)/2 reflection
// coefficients found so far, so every pass over the window recomputes it from the track point
// (two loads and ~30 flops per row).  Nothing per-instance lives in memory but the track, which
// all instances share read-only.  Written as plain inline code so that tests can compile the same
// source for the CPU (tests/emu/) and compare it with the oracle without a GPU.
#pragma once

#ifndef NMPC_DEV
#define NMPC_DEV inline
#endif

#include <math.h>

namespace nmpcf {

constexpr double MAX_LEN = 15.0;     // src/PathTrackingSys.cpp:108

// getTrackSection: strict '<' keeps the first minimum; end - start >= 4; wrap to [0,4) at the end
NMPC_DEV void track_section(const double *xy, int n, double sx, double sy, int *start, int *end)
{
    double best = (sx - xy[0]) * (sx - xy[0]) + (sy - xy[1]) * (sy - xy[1]);
    int mi = 0;
    for (int i = 1; i < n; ++i) {
        const double dx = sx - xy[2 * i], dy = sy - xy[2 * i + 1];
        const double d = dx * dx + dy * dy;
        if (d < best) { best = d; mi = i; }
    }
    double len = 0.0;
    int fi = mi;
    while (len < MAX_LEN && fi < n - 1) {
        ++fi;
        const double dx = xy[2 * fi] - xy[2 * (fi - 1)], dy = xy[2 * fi + 1] - xy[2 * (fi - 1) + 1];
        len += sqrt(dx * dx + dy * dy);
    }
    int s = mi, e = fi;
    if (e - s < 4) e = s + 4;
    if (e >= n) { s = 0; e = s + 4; }
    *start = s;
    *end = e;
}

struct Fit {
    double f[4][5];     // f[s][c], c > s: reflection coefficients 2 v's a_c / v'v of step s
    double R[4][5];     // finished rows of [R | Q'y]
};

// state of row i (original [1, x, x^2, x^3, y]) after the first j reflections, i >= j
NMPC_DEV void row_state(const Fit &F, int j, double x, double y, double *a)
{
    a[0] = 1.0; a[1] = x; a[2] = a[1] * x; a[3] = a[2] * x; a[4] = y;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < j) {
#pragma unroll
            for (int c = s + 1; c < 5; ++c) a[c] -= F.f[s][c] * a[s];
        }
}

// cubic least squares over track points [start, end) seen from (sx, sy) in a frame rotated by
// `angle` (interpolate: angle = rotation - psi); returns the sum of squared residuals
NMPC_DEV double fit_rotation(const double *xy, int start, int end, double sx, double sy, double angle,
                             double *coeffs)
{
    const double ca = cos(angle), sa = sin(angle);
    const int n = end - start;
    Fit F;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 5; ++c) { F.f[s][c] = 0.0; F.R[s][c] = 0.0; }
    const int steps = n < 4 ? n : 4;
    // the four elimination steps are written out (full unroll): every index into F is then a
    // compile-time constant and F lives in registers -- indexed by a loop variable it went to scratch
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= steps) break;
        // pass 1: column norm below and including the pivot
        double nrm2 = 0.0, below2 = 0.0, piv[5] = {0, 0, 0, 0, 0};
#pragma unroll 1
        for (int i = j; i < n; ++i) {
            const double dx = xy[2 * (start + i)] - sx, dy = xy[2 * (start + i) + 1] - sy;
            double a[5];
            row_state(F, j, dx * ca - dy * sa, dx * sa + dy * ca, a);
            const double aj = j == 0 ? a[0] : j == 1 ? a[1] : j == 2 ? a[2] : a[3];
            nrm2 += aj * aj;
            if (i == j) {
#pragma unroll
                for (int c = 0; c < 5; ++c) piv[c] = a[c];
            } else {
                below2 += aj * aj;
            }
        }
        const double nrm = sqrt(nrm2);
        const double pj = j == 0 ? piv[0] : j == 1 ? piv[1] : j == 2 ? piv[2] : piv[3];
        if (nrm == 0.0) {      // zero column: nothing to reflect, the pivot row stays as it is
#pragma unroll
            for (int c = 0; c < 5; ++c) if (c >= j) F.R[j][c] = piv[c];
            continue;
        }
        const double alpha = pj > 0.0 ? -nrm : nrm;
        const double v0 = pj - alpha;
        const double vnorm2 = v0 * v0 + below2;
        if (vnorm2 == 0.0) {
#pragma unroll
            for (int c = 0; c < 5; ++c) if (c >= j) F.R[j][c] = piv[c];
            continue;
        }
        // pass 2: v'a_c for the remaining columns
        double dot[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 5; ++c) dot[c] = v0 * piv[c];
#pragma unroll 1
        for (int i = j + 1; i < n; ++i) {
            const double dx = xy[2 * (start + i)] - sx, dy = xy[2 * (start + i) + 1] - sy;
            double a[5];
            row_state(F, j, dx * ca - dy * sa, dx * sa + dy * ca, a);
            const double aj = j == 0 ? a[0] : j == 1 ? a[1] : j == 2 ? a[2] : a[3];
#pragma unroll
            for (int c = 0; c < 5; ++c) dot[c] += aj * a[c];
        }
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (c > j) {
                const double fc = 2.0 * dot[c] / vnorm2;
                F.f[j][c] = fc;
                F.R[j][c] = piv[c] - fc * v0;
            }
        F.R[j][j] = alpha;
        // rows below the pivot row see the reflection through a_c -= f a_j; the pivot row's own
        // entries are final now.  For the recursion of later rows the pivot row is not needed.
    }
    // back substitution on the 4x4 triangle
    double cf[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        double acc = F.R[j][4];
#pragma unroll
        for (int c = j + 1; c < 4; ++c) acc -= F.R[j][c] * cf[c];
        cf[j] = acc / F.R[j][j];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) coeffs[c] = cf[c];
    double cost = 0.0;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        const double dx = xy[2 * (start + i)] - sx, dy = xy[2 * (start + i) + 1] - sy;
        const double x = dx * ca - dy * sa, y = dx * sa + dy * ca;
        const double e = y - (cf[0] + cf[1] * x + cf[2] * x * x + cf[3] * x * x * x);
        cost += e * e;
    }
    return cost;
}

// calcCoeffs + the frame transform of PathTrackingSys::solve(state, pitch, vRef):
// ws = world state [x,y,psi,v,delta,thr]; outputs the solver-frame state and parameters
NMPC_DEV void frontend(const double *xy, int n, const double *ws, double pitch, double vref,
                       double *x0s, double *par, double *rotation)
{
    int start, end;
    track_section(xy, n, ws[0], ws[1], &start, &end);
    double minCost = 1e19, rot_best = 0.0, cbest[4] = {0, 0, 0, 0};
    // the reference's loop `for (rot = -pi/2; rot < 0; rot += pi/6)` runs 4 times in fp64
    double rot = -1.57079632679489661923;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        double cur[4];
        const double cost = fit_rotation(xy, start, end, ws[0], ws[1], rot - ws[2], cur);
        if (cost < minCost) {
            minCost = cost;
            rot_best = rot;
#pragma unroll
            for (int c = 0; c < 4; ++c) cbest[c] = cur[c];
        }
        rot += 1.57079632679489661923 / 3;
    }
    x0s[0] = 0.0; x0s[1] = 0.0; x0s[2] = rot_best; x0s[3] = ws[3]; x0s[4] = ws[4]; x0s[5] = ws[5];
    par[0] = cbest[0]; par[1] = cbest[1]; par[2] = cbest[2]; par[3] = cbest[3]; par[4] = pitch; par[5] = vref;
    *rotation = rot_best;
}

// un-rotation of a predicted point (src/PathTrackingSys.cpp:27-41; psi is NOT un-rotated)
NMPC_DEV void unrotate(const double *ws, double rotation, double lx, double ly, double *wx, double *wy)
{
    const double a = ws[2] - rotation;
    const double s = sin(a), c = cos(a);
    *wx = lx * c - ly * s + ws[0];
    *wy = lx * s + ly * c + ws[1];
}

// Closed-loop plant: classical RK4 of (X,Y,psi,v) over `dtp` seconds with
// the commanded steering angle and throttle held, then the reference's feedback rule
// (src/RosMpc.cpp:81-90): delta <- (x1.delta * ratio) / ratio, throttle <- x1.throttle.
// measured != 0 is the other branch of that rule (use_actual_steering_topic = true, src/RosMpc.cpp:88-90,150):
// the steering angle of the state comes from a measurement, not from the command -- the car moves with the
// angle it has (ws[4], set from the measurement before the solve) and that entry is left alone.
NMPC_DEV void plant_step(double *ws, double delta_cmd, double thr_cmd, double pitch, double dtp,
                         double inv_wheelbase, double steering_ratio, int measured = 0)
{
    const double delta = measured ? ws[4] : (delta_cmd * steering_ratio) / steering_ratio;
    const double td = tan(delta) * inv_wheelbase, gp = sin(pitch) * 9.81;
    double s[4] = {ws[0], ws[1], ws[2], ws[3]}, acc[4] = {0, 0, 0, 0}, k[4] = {0, 0, 0, 0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const double c = st == 0 ? 0.0 : st == 3 ? dtp : 0.5 * dtp;
        const double psi = s[2] + c * k[2], v = s[3] + c * k[3];
        k[0] = v * cos(psi);
        k[1] = v * sin(psi);
        k[2] = v * td;
        k[3] = 5.0 * thr_cmd - 0.087 * v + gp;
        const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] += wgt * k[m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) ws[m] = s[m] + dtp / 6.0 * acc[m];
    ws[4] = delta;
    ws[5] = thr_cmd;
}

}  // namespace nmpcf
