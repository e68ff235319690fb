"""nmpc_frontend.h (the per-thread GPU front-end, compiled for the CPU) against the oracle's
restatement of PathTrackingSys::calcCoeffs / getTrackSection / util::polyfit."""
import math

import numpy as np

import emu_binding as emu
from oracle import oracle as orc


def world_states(B, seed=3):
    """Cars near the reference's test track, headed roughly along it, with lateral/heading offsets.
    They stay clear of the seam: track[400] duplicates track[0] to ~1e-14, so there the reference's
    strict-'<' nearest-point search (PathTrackingSys.cpp:110-117) is decided by round-off and flips
    between the 15 m window at index 0 and the 4-point wrap window at index 400."""
    t = orc.test_track()
    rng = np.random.default_rng(seed)
    idx = rng.integers(8, 388, size=B)
    d = t[idx + 1] - t[idx]
    heading = np.arctan2(d[:, 1], d[:, 0])
    nrm = np.stack([-np.sin(heading), np.cos(heading)], axis=1)
    ws = np.zeros((B, 6))
    ws[:, 0:2] = t[idx] + nrm * rng.normal(0, 0.7, size=(B, 1)) + rng.normal(0, 0.1, size=(B, 2))
    ws[:, 2] = heading + rng.normal(0, 0.15, size=B)
    ws[:, 3] = rng.uniform(0, 8, size=B)
    ws[:, 4] = rng.uniform(-0.3, 0.3, size=B)
    ws[:, 5] = rng.uniform(-0.5, 0.8, size=B)
    pv = np.stack([rng.uniform(-0.05, 0.05, size=B), rng.uniform(3, 7, size=B)], axis=1)
    return t, ws, pv


def test_frontend_matches_oracle():
    t, ws, pv = world_states(300)
    ws[0] = [41.0, 0.0, math.pi / 2, 5.0, 0.0, 0.0]          # tests/testMpc.h:13
    ws[1, 0:2] = t[399]                                       # end-of-track wrap rule
    ws[2, 0:2] = t[398] + 0.01
    x0s, par, rot = emu.frontend(t, ws, pv)
    nrot = 0
    for i in range(ws.shape[0]):
        r, c = orc.calc_coeffs(t, ws[i])
        st, en = orc.track_section(t, ws[i, 0], ws[i, 1])
        if en - st == 4:
            # a cubic through 4 points is exact in every frame: all four costs are round-off and
            # the arg-min is arbitrary (on both sides); only the frame-independent outputs compare
            assert np.array_equal(x0s[i, 3:], ws[i, 3:]) and np.array_equal(par[i, 4:], pv[i])
            continue
        assert rot[i] == r, (i, rot[i], r)
        nrot += r != -math.pi / 2
        assert np.abs(par[i, :4] - c).max() <= 1e-9 * max(1.0, np.abs(c).max()), (i, par[i, :4], c)
        assert np.array_equal(x0s[i], [0.0, 0.0, r, ws[i, 3], ws[i, 4], ws[i, 5]])   # PathTrackingSys.cpp:23
        assert np.array_equal(par[i, 4:], pv[i])
    assert nrot > 10      # the cases do exercise several of the four rotations


def test_plant_step_is_rk4_of_the_model():
    cfg = orc.default_config(40, 1.0 / 30.0)        # the oracle's RK4 map with dt = the tick
    rng = np.random.default_rng(1)
    for _ in range(20):
        ws = np.array([rng.normal(0, 10), rng.normal(0, 10), rng.uniform(-3, 3), rng.uniform(0, 8), 0.1, 0.2])
        dcmd, tcmd, pitch = rng.uniform(-0.5, 0.5), rng.uniform(-1, 1), rng.uniform(-0.05, 0.05)
        out = emu.plant_step(ws, dcmd, tcmd, pitch, 1.0 / 30.0)
        x = ws.copy(); x[4] = dcmd; x[5] = tcmd
        phi, _, _ = orc.rk4_sens(cfg, x, np.zeros(2), np.array([0, 0, 0, 0, pitch, 0.0]))
        assert np.abs(out[:4] - phi[:4]).max() <= 1e-12 * max(1.0, np.abs(phi).max())
        assert out[4] == dcmd and out[5] == tcmd
