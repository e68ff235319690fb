"""The kernel body (mpc_local_planner_amd/csrc/nmpc_wave_solver.h), run on the CPU lane emulator
(tests/emu/), against the oracle: checks lane mappings, the LDS layout and the exchange protocol
on a machine without a GPU.  The GPU itself is checked in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import emu_binding as emu
from mpc_local_planner_amd import scenarios as sc
from oracle import oracle as orc

TOL = 1e-9


def _cfg(N, kernel="wave", **over):
    """The four-slot kernel runs the interior-point iteration to its own termination test (no active-set
    polish): its oracle counterpart is the same configuration with the polish off."""
    return orc.default_config(N, 0.35, **(dict(over, polish=0) if kernel == "quad" else over))


def _err(g, xo, uo):
    B = xo.shape[0]
    return np.maximum(np.abs(g["x"] - xo).reshape(B, -1).max(1), np.abs(g["u"] - uo).reshape(B, -1).max(1))


@pytest.mark.parametrize("N,gen,kernel", [(40, sc.config2, "quad"), (10, sc.config2, "quad"), (63, sc.config2, "quad"),
                                          (1, sc.config2, "quad"), (2, sc.config2, "quad"),
                                          (40, sc.config2, "wave"), (1, sc.config2, "wave"),
                                          (40, sc.config4, "wave"), (64, sc.config2, "wave"), (100, sc.config4, "wave"),
                                          (63, sc.config2, "wave"), (50, sc.config4, "wave")])
def test_emulated_wave_matches_oracle_cold_and_warm(N, gen, kernel):
    """Both solve kernels: "quad" = four instances per workgroup (SQP-RTI up to N = 63, what nmpc_solve
    launches there), "wave" = one instance per workgroup (SQP mode and longer horizons)."""
    B = 5 if N <= 63 else 3             # (not a multiple of four: three slots of the workgroup run empty at the end)
    cfg = _cfg(N, kernel)
    x0, p = gen(B)
    g = emu.solve_batch(cfg, x0, p, kernel=kernel)
    xo, uo, so, io = orc.rti_batch(cfg, x0, p)
    assert (g["status"] == so).all() and (so == 0).all()
    assert (np.abs(g["ipm_iters"] - io) <= 1).all()
    scale = max(1.0, np.abs(xo).max())
    assert _err(g, xo, uo).max() <= 1e-6 * scale
    assert np.median(_err(g, xo, uo)) <= TOL * scale
    # control record layout (include/nmpc.h: nmpc_get_controls)
    assert np.array_equal(g["ctrl"][:, 0:2], g["u"][:, 0, :])
    if N >= 2:
        assert np.array_equal(g["ctrl"][:, 2:4], g["x"][:, 1, 4:6])
    # warm call from the oracle's iterate (no shift)
    traj = np.zeros((B, N + 1, 8)); traj[:, :, :6] = xo; traj[:, :N, 6:] = uo
    g2 = emu.solve_batch(cfg, x0, p, traj=traj, cold=False, kernel=kernel, sweeper=2 if kernel == "quad" else 0)
    xo2, uo2, so2, _ = orc.rti_batch(cfg, x0, p, xo.copy(), uo.copy(), cold=False)
    ok = so2 == 0
    assert (g2["status"][ok] == 0).all()
    assert _err(g2, xo2, uo2)[ok].max() <= 1e-6 * max(1.0, np.abs(xo2).max())


@pytest.mark.parametrize("kernel", ["wave", "quad"])
def test_emulated_wave_per_instance_weights_and_bounds(kernel):
    N, B = 20, 6
    cfg = _cfg(N, kernel)
    x0, p = sc.config2(B)
    rng = np.random.default_rng(3)
    W = np.array(cfg.W[:])[None, :] * rng.uniform(0.5, 2.0, size=(B, 7))
    bnd = np.tile(np.array([-80, -0.33, -0.57, -1.0, 80, 0.33, 0.57, 1.0]), (B, 1))
    bnd[:, 1] *= rng.uniform(0.5, 1.5, size=B); bnd[:, 5] = -bnd[:, 1]
    g = emu.solve_batch(cfg, x0, p, W=W, bnd=bnd, kernel=kernel)
    for i in range(B):
        oc = _cfg(N, kernel)
        for m in range(7):
            oc.W[m] = W[i, m]
        oc.lbu[1], oc.ubu[1] = bnd[i, 1], bnd[i, 5]
        xo, uo = orc.init_guess(oc, x0[i])
        st = orc.rti(oc, x0[i], p[i], xo, uo)
        assert st.status == 0 and g["status"][i] == 0
        assert np.abs(g["x"][i] - xo).max() <= 1e-7 and np.abs(g["u"][i] - uo).max() <= 1e-7
        assert abs(g["cost"][i] - st.cost) <= 1e-9 * max(1.0, abs(st.cost))


@pytest.mark.parametrize("N,B,gen", [(30, 3, sc.config2), (70, 1, sc.config4)])
def test_emulated_wave_sqp_mode_converges_to_the_oracle_fixed_point(N, B, gen):
    """N = 70: two waves per instance with idle threads behind node N (x_{k+1} and the multipliers of
    the last QP then travel through the LDS image instead of lane shuffles)."""
    cfg = orc.default_config(N, 0.35)
    x0, p = gen(B)
    g = emu.solve_batch(cfg, x0, p, sqp_max_iter=30, sqp_tol=1e-9)
    for i in range(B):
        xo, uo = orc.init_guess(cfg, x0[i])
        st = orc.sqp(cfg, x0[i], p[i], xo, uo, 30, 1e-9)
        assert st.status == 0 and g["status"][i] == 0
        assert np.abs(g["x"][i] - xo).max() <= 1e-6 and np.abs(g["u"][i] - uo).max() <= 1e-6
        # same termination test (KKT residuals of the NLP with the last QP's multipliers)
        assert g["sqp_iters"][i] == st.sqp_iters >= 2
        assert abs(g["cost"][i] - st.cost) <= 1e-9 * max(1.0, abs(st.cost))


@pytest.mark.parametrize("N,B", [(20, 2), (100, 1)])
def test_emulated_wave_sqp_mode_iteration_limit(N, B):
    """Status 2 (acados: maximum iterations) after exactly max_iter QPs when the KKT test still
    fails; the iterate equals the oracle's after the same number of full steps.  N = 100: two waves
    and the 48-double stage layout (nmpcw::compact_layout), multipliers and costates through LDS."""
    cfg = orc.default_config(N, 0.35)
    x0, p = sc.config2(B)
    g = emu.solve_batch(cfg, x0, p, sqp_max_iter=2, sqp_tol=1e-12)
    for i in range(B):
        xo, uo = orc.init_guess(cfg, x0[i])
        st = orc.sqp(cfg, x0[i], p[i], xo, uo, 2, 1e-12)
        assert st.status == 2 and g["status"][i] == 2
        assert g["sqp_iters"][i] == st.sqp_iters == 2
        assert np.abs(g["x"][i] - xo).max() <= 1e-7 and np.abs(g["u"][i] - uo).max() <= 1e-7


@pytest.mark.parametrize("kernel", ["wave", "quad"])
def test_emulated_wave_failure_paths_match_the_oracle(kernel):
    """Status 1 (NaN in the measurement: the iterate is left as it was) and status 4 (interior-point
    iteration limit: the step is still taken, as acados returns the failed trajectory,
    src/AcadosSolver.cpp:56-59) -- same status, same iterate as the oracle; other instances of the
    batch are not affected."""
    N, B = 20, 4
    cfg = _cfg(N, kernel)
    x0, p = sc.config2(B)
    x0[1, 3] = np.nan
    g = emu.solve_batch(cfg, x0, p, kernel=kernel)
    xo, uo, so, io = orc.rti_batch(cfg, x0, p)
    assert g["status"].tolist() == so.tolist() == [0, 1, 0, 0]
    ok = so == 0
    assert _err(dict(x=g["x"][ok], u=g["u"][ok]), xo[ok], uo[ok]).max() <= 1e-7
    x0, p = sc.config2(B)
    cfg3 = _cfg(N, kernel, ipm_max_iter=3)
    g = emu.solve_batch(cfg3, x0, p, kernel=kernel)
    xo, uo, so, io = orc.rti_batch(cfg3, x0, p)
    assert (g["status"] == 4).all() and (so == 4).all() and (g["ipm_iters"] == 3).all()
    assert _err(g, xo, uo).max() <= 1e-9


@pytest.mark.parametrize("kernel", ["quad", "wave"])
def test_emulated_kernels_on_an_ill_conditioned_stage_cost(kernel):
    """A path polynomial with curvature coefficient c2 = 170 (found by tools/stress_parity.py: 1 % of an
    extreme neighbour mixed into the parameters) makes the stage costs ill-conditioned.  The four-slot
    kernel forms the rows of the cost-to-go independently; without symmetrising it every few stages the
    unsymmetric round-off grew along the factor sweep until the QP 'converged' 7e-2 away from the solution
    (36 interior-point iterations instead of 15 at N = 63).  Both kernels must stay at the oracle's result."""
    xa, pa = sc.config2(2, start=20_000_000 + 16767)
    x1 = (xa[1] + 0.01 * (xa[0] - xa[1]))[None]
    p1 = (pa[1] + 0.01 * (pa[0] - pa[1]))[None]
    assert 150.0 < p1[0, 2] < 200.0
    for N in (63,):
        cfg = _cfg(N, kernel)
        xo, uo, so, _ = orc.rti_batch(cfg, xa[1:2], pa[1:2])
        xo2, uo2, so2, io2 = orc.rti_batch(cfg, x1, p1, xo.copy(), uo.copy(), cold=False)
        assert so2[0] == 0
        traj = np.zeros((1, N + 1, 8)); traj[:, :, :6] = xo; traj[:, :N, 6:] = uo
        g = emu.solve_batch(cfg, x1, p1, traj=traj, cold=False, kernel=kernel)
        assert g["status"][0] == 0 and abs(int(g["ipm_iters"][0]) - int(io2[0])) <= 1
        assert _err(g, xo2, uo2).max() <= 1e-7 * max(1.0, np.abs(xo2).max())
