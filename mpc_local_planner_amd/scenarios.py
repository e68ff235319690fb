"""Deterministic synthetic inputs for the batched path-tracking solver.

Counter-based RNG (splitmix64): draw j of instance i depends only on
(seed, i, j), so any shard of a batch can be generated independently on any
rank.  Distributions follow SURVEY.md section 8(d) (configs 2/3 and 4): they
mimic what the reference's path front-end hands the solver -- the car at the
origin of a rotated frame (PathTrackingSys.cpp:23) and a cubic fitted to the
next ~15 m of path (PathTrackingSys.cpp:63-105).
"""
import numpy as np

SEED = 0x5EED0001
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed, idx, draw):
    """U[0,1) for instance indices `idx` (array) and draw number `draw`."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _mix(idx + np.uint64(seed))
        v = _mix(key + np.uint64(draw) * _GOLD)
    return (v >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, idx, draw, lo, hi):
    return lo + (hi - lo) * uniform01(seed, idx, draw)


def normal(seed, idx, draw, sigma):
    """N(0, sigma) by Box-Muller on draws (draw, draw+1)."""
    u1 = 1.0 - uniform01(seed, idx, draw)  # (0,1]
    u2 = uniform01(seed, idx, draw + 1)
    return sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _pack(psi0, v, delta, thr, pitch, epsi0, cte0, kappa, c3, vref):
    n = psi0.shape[0]
    s0 = np.tan(psi0 - epsi0)
    x0 = np.zeros((n, 6))
    x0[:, 2], x0[:, 3], x0[:, 4], x0[:, 5] = psi0, v, delta, thr
    p = np.zeros((n, 6))
    p[:, 0] = cte0
    p[:, 1] = s0
    p[:, 2] = kappa * (1.0 + s0 * s0) ** 1.5 / 2.0
    p[:, 3] = c3
    p[:, 4] = pitch
    p[:, 5] = vref
    return x0, p


def config2(B, start=0, seed=SEED, vref=5.0):
    """Solver-frame states x0[B,6] and parameters p[B,6] for configs 2/3
    (instances start .. start+B-1 of the global batch)."""
    i = np.arange(start, start + B, dtype=np.uint64)
    psi0 = uniform(seed, i, 0, -1.2, 0.2)
    v = uniform(seed, i, 1, 0.0, 8.0)
    delta = uniform(seed, i, 2, -0.3, 0.3)
    thr = uniform(seed, i, 3, -0.5, 0.8)
    pitch = uniform(seed, i, 4, -0.05, 0.05)
    epsi0 = normal(seed, i, 5, 0.15)
    cte0 = normal(seed, i, 7, 0.5)
    kappa = uniform(seed, i, 9, -0.05, 0.05)
    c3 = uniform(seed, i, 10, -0.002, 0.002)
    return _pack(psi0, v, delta, thr, pitch, epsi0, cte0, kappa, c3, vref)


def config4(B, start=0, seed=SEED, vref=5.0):
    """Hard instances for the full-SQP config: large offsets, steering/throttle
    close to their bounds so that constraints are active."""
    i = np.arange(start, start + B, dtype=np.uint64)
    sgn = lambda d: np.where(uniform01(seed, i, d) < 0.5, -1.0, 1.0)
    # psi0 range narrowed w.r.t. config 2 so that the path heading psi0 - epsi0 stays inside
    # (-1.3, 0.9): the reference's front-end never hands the solver a near-vertical cubic
    # (it picks the frame rotation with the best fit, PathTrackingSys.cpp:63-79)
    psi0 = uniform(seed, i, 0, -0.6, 0.2)
    v = uniform(seed, i, 1, 0.0, 2.0)
    delta = sgn(11) * uniform(seed, i, 2, 0.45, 0.57)
    thr = uniform(seed, i, 3, 0.8, 1.0)
    pitch = uniform(seed, i, 4, -0.05, 0.05)
    epsi0 = sgn(12) * uniform(seed, i, 5, 0.4, 0.7)
    cte0 = sgn(13) * uniform(seed, i, 7, 2.0, 4.0)
    kappa = uniform(seed, i, 9, -0.05, 0.05)
    c3 = uniform(seed, i, 10, -0.002, 0.002)
    return _pack(psi0, v, delta, thr, pitch, epsi0, cte0, kappa, c3, vref)


def test_track():
    """The reference's demo track (util::getTestTrack, src/utilities.cpp:59-72) as input data:
    theta accumulates 2*pi/400 in fp64 until it reaches 2*pi, which yields 401 points, the last a
    near-duplicate of the first."""
    import math
    pts = []
    theta, step = 0.0, 2 * math.pi / 400.0
    while theta < 2 * math.pi:
        pts.append((2 * 20 * math.cos(theta), 20 * math.sin(1 * theta) + 20 / 1.2 * math.sin(3 * theta)))
        theta += step
    return np.array(pts)


def cars_on_track(B, track, seed=5, lateral=0.3, heading=0.1, v=(2.0, 6.0), vref=(3.0, 7.0)):
    """World-frame states near `track` heading along it (config 5), clear of the seam where the
    last track point duplicates the first, and (pitch, v_ref) per vehicle."""
    rng = np.random.default_rng(seed)
    idx = rng.integers(8, track.shape[0] - 13, size=B)
    d = track[idx + 1] - track[idx]
    hd = np.arctan2(d[:, 1], d[:, 0])
    ws = np.zeros((B, 6))
    ws[:, 0:2] = track[idx] + rng.normal(0, lateral, size=(B, 2))
    ws[:, 2] = hd + rng.normal(0, heading, size=B)
    ws[:, 3] = rng.uniform(v[0], v[1], size=B)
    pv = np.stack([np.zeros(B), rng.uniform(vref[0], vref[1], size=B)], axis=1)
    return ws, pv
