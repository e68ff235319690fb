"""Oracle restatement of the path front-end and BASELINE config 1
(tests/testMpc.h:13 initial state on util::getTestTrack), CPU only."""
import math

import numpy as np

from oracle import oracle as orc

STATE1 = np.array([41.0, 0.0, math.pi / 2, 5.0, 0.0, 0.0])     # tests/testMpc.h:13


def test_test_track_has_401_points():
    """src/utilities.cpp:59-72: the fp accumulation of theta yields 401 points, the last one a
    near-duplicate of the first."""
    t = orc.test_track()
    assert t.shape == (401, 2)
    assert np.abs(t[0] - [40.0, 0.0]).max() < 1e-12
    assert np.abs(t[400] - t[0]).max() < 1e-9
    th = 2 * math.pi / 400 * 100
    assert np.abs(t[100] - [40 * math.cos(th), 20 * math.sin(th) + 20 / 1.2 * math.sin(3 * th)]).max() < 1e-9


def test_rotation_loop_has_four_values():
    """src/PathTrackingSys.cpp:69: for (rot = -pi/2; rot < 0; rot += pi/6) runs 4 times in fp64."""
    rots, rot = [], -math.pi / 2
    while rot < 0:
        rots.append(rot)
        rot += (math.pi / 2) / 3
    assert len(rots) == 4 and rots[3] == -2.220446049250313e-16


def test_polyfit_matches_least_squares():
    rng = np.random.default_rng(5)
    for n in (4, 5, 17, 60):
        xs = np.sort(rng.uniform(-2, 15, n))
        ys = 0.3 - 0.2 * xs + 0.05 * xs ** 2 - 0.001 * xs ** 3 + rng.normal(0, 0.01, n)
        c = orc.polyfit3(xs, ys)
        V = np.vander(xs, 4, increasing=True)
        ref = np.linalg.lstsq(V, ys, rcond=None)[0]
        assert np.abs(c - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())


def test_track_section_rules():
    t = orc.test_track()
    s, e = orc.track_section(t, 41.0, 0.0)
    assert s in (0, 400) and e - s >= 4
    seg = np.sqrt(((t[s + 1:e + 1] - t[s:e]) ** 2).sum(1)).sum()
    assert 15.0 <= seg < 15.0 + 1.5
    # near the end of the track the reference wraps to the first 4 points (PathTrackingSys.cpp:130-133)
    s2, e2 = orc.track_section(t, *t[399])
    assert (s2, e2) in ((0, 4), (399, 400)) or e2 < 401


def test_config1_front_end_and_single_rti():
    t = orc.test_track()
    rot, c = orc.calc_coeffs(t, STATE1)
    assert any(abs(rot - r) < 1e-15 for r in (-math.pi / 2, -math.pi / 3, -math.pi / 6, -2.220446049250313e-16))
    # the car starts 1 m beside the track point (40, 0): the fitted cubic passes about 1 m away
    assert 0.5 < abs(c[0]) < 2.0
    for N in (30, 40):
        cfg = orc.default_config(N, 0.35)
        xs = np.array([0.0, 0.0, rot, STATE1[3], STATE1[4], STATE1[5]])    # PathTrackingSys.cpp:23
        p = np.array([c[0], c[1], c[2], c[3], 0.0, 5.0])
        xt, ut = orc.init_guess(cfg, xs)
        st = orc.rti(cfg, xs, p, xt, ut)
        assert st.status == 0
        assert np.array_equal(xt[0], xs)
        assert np.isfinite(xt).all() and np.isfinite(ut).all()
        # the predicted path closes in on the cubic inside the fitted 15 m window (beyond it the
        # cubic is an extrapolation, as in the reference)
        X = xt[:, 0]
        cte = c[0] + c[1] * X + c[2] * X ** 2 + c[3] * X ** 3 - xt[:, 1]
        assert np.abs(cte[:6]).max() < 1.3 and abs(cte[4]) < 0.5 * abs(cte[0])
