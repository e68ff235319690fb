"""The CPU oracle against the committed high-precision fixtures (tests/golden/make_golden.py).

The reference holds no golden vectors for this path: acados parity is
unpinned, and these fixtures pin the oracle against an independent 50-digit formulation of the
model and cost the reference states in acados/mpcPathTracking.py.
"""
import os

import numpy as np

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-13


def _close(a, b, scale=None):
    s = max(1.0, np.abs(b).max()) if scale is None else scale
    return np.abs(a - b).max() <= RTOL * s


def test_rk4_map_and_sensitivities():
    g = np.load(os.path.join(GOLD, "rk4_map.npz"))
    cfg = orc.default_config(40, float(g["dt"]))
    assert cfg.wheelbase == float(g["wheelbase"])
    for i in range(g["x"].shape[0]):
        phi, A, B = orc.rk4_sens(cfg, g["x"][i], g["u"][i], g["p"][i])
        assert _close(phi, g["phi"][i]), i
        assert _close(A, g["A"][i]), (i, np.abs(A - g["A"][i]).max())
        assert _close(B, g["B"][i]), (i, np.abs(B - g["B"][i]).max())


def test_rk4_structure_the_kernel_relies_on():
    """Columns X,Y of A are unit vectors, rows delta,thr are exact integrators, the v-row does
    not depend on the state (nmpc_wave_solver.h keeps those entries as launch constants)."""
    g = np.load(os.path.join(GOLD, "rk4_map.npz"))
    A, B = g["A"], g["B"]
    dt = float(g["dt"])
    eye = np.eye(6)
    assert np.abs(A[:, :, 0] - eye[:, 0]).max() == 0 and np.abs(A[:, :, 1] - eye[:, 1]).max() == 0
    assert np.abs(A[:, 4, :] - eye[4]).max() < 1e-15 and np.abs(A[:, 5, :] - eye[5]).max() < 1e-15
    assert np.abs(B[:, 4, 0] - dt).max() < 1e-15 and np.abs(B[:, 5, 1] - dt).max() < 1e-15
    assert np.abs(A[:, 3, :] - A[0, 3, :]).max() < 1e-15 and np.abs(B[:, 3, :] - B[0, 3, :]).max() < 1e-15
    assert np.abs(A[:, 2, 2] - 1.0).max() < 1e-15


def test_nls_cost_terms():
    g = np.load(os.path.join(GOLD, "nls_cost.npz"))
    cfg = orc.default_config(40, float(g["dt"]))
    assert np.array_equal(np.array(cfg.W[:]), g["W"])
    for i in range(g["x"].shape[0]):
        y, J = orc.cost_yJ(cfg, g["x"][i], g["u"][i], g["p"][i])
        H, gr = orc.stage_gn(cfg, g["x"][i], g["u"][i], g["p"][i])
        assert _close(y, g["y"][i]) and _close(J, g["J"][i]), i
        assert _close(H, g["H"][i]) and _close(gr, g["g"][i]), i


def test_reference_defaults():
    """config/mpc_params.yaml:35-59 and acados/mpcPathTracking.py:103."""
    c = orc.default_config(30, 0.35)
    assert (c.N, c.dt, c.wheelbase) == (30, 0.35, 2.65)
    assert list(c.W) == [5, 5, 10, 0.01, 0.1, 0.5, 0.1]
    assert (c.lbu[0], c.ubu[0], c.lbu[1], c.ubu[1]) == (-80.0, 80.0, -0.33, 0.33)
    assert (c.lbx[0], c.ubx[0], c.lbx[1], c.ubx[1]) == (-0.57, 0.57, -1.0, 1.0)
