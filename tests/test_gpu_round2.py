"""GPU tests added in round 2 (each through the C ABI, against the oracle): free-running long-horizon
RTI chains, zero-copy device inputs, steering ratio in the closed loop, config 5 at full size,
status-2 restart, argument validation, and the DPP canary."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
from test_frontend_emu import world_states
from test_gpu_frontend_closed_loop import _oracle_closed_loop
from test_gpu_parity import TOL_BAR, _err, _ocfg, _scale, assert_parity

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("N", [100, 127])
def test_free_running_rti_chain_long_horizon(N):
    """1 cold + 4 warm SQP-RTI calls at N = 100 / 127 WITHOUT re-seeding the kernel from the oracle's
    iterate between calls (test_cold_then_warm_rti_matches_oracle re-seeds beyond N = 40): every instance
    stays within the 1e-6 bar on every call.  Measured drift (tools/drift_probe.py, 64 instances, six
    calls): worst instance 3e-10 -> 6e-8 at N = 100, 7e-12 -> 5e-9 at N = 127, median <= 3e-12."""
    B = 64
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0)
    s.set_params(p)
    ocfg = _ocfg(N)
    xo = uo = None
    for call in range(5):
        s.solve(B)
        g = s.get(B)
        xo, uo, so, io = orc.rti_batch(ocfg, x0, p, xo, uo, cold=(call == 0))
        assert (so == 0).all() and (g["status"] == 0).all()
        e = _err(g, xo, uo) / _scale(xo, uo)
        assert e.max() <= TOL_BAR, (call, e.max())
        assert np.median(e) <= 1e-10, (call, np.median(e))
        if call == 0:
            assert e.max() <= 1e-9
    s.close()


def test_device_resident_inputs_match_host_inputs():
    """nmpc_set_inputs_device (zero-copy x0 / p in the caller's device memory) gives exactly what
    nmpc_set_x0 / nmpc_set_params give, for a cold and a warm call; switching back to host inputs works."""
    import torch
    N, B = 40, 300
    x0, p = sc.config2(B)
    x0b, pb = sc.config2(B, start=7000)
    x0b, pb = 0.9 * x0 + 0.1 * x0b, 0.9 * p + 0.1 * pb
    a = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    b = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    dev = torch.device("cuda:0")
    for xa, pa in ((x0, p), (x0b, pb)):
        a.set_x0(xa)
        a.set_params(pa)
        a.solve(B)
        ga = a.get(B)
        dx, dp = torch.as_tensor(xa).to(dev), torch.as_tensor(pa).to(dev)
        torch.cuda.synchronize()
        b.set_inputs_device(C.c_void_p(dx.data_ptr()), C.c_void_p(dp.data_ptr()), B)
        b.solve(B)
        gb = b.get(B)
        for k in ("x", "u", "status", "ipm_iters", "cost"):
            assert np.array_equal(ga[k], gb[k]), k
    b.set_x0(x0)          # host inputs again: the handle's own buffers are used
    b.set_params(p)
    a.set_x0(x0)
    a.set_params(p)
    a.solve(B)
    b.solve(B)
    assert np.array_equal(a.get(B)["x"], b.get(B)["x"])
    xo, uo, _, _ = orc.rti_batch(_ocfg(N), x0, p)      # and the first call is the oracle's cold solve
    a.reset(None, B)
    a.solve(B)
    assert_parity(a.get(B), xo, uo)
    a.close()
    b.close()


def test_closed_loop_with_the_reference_steering_ratio():
    """steering_ratio = 17.3 (config/mpc_params.yaml:62): the command is published as delta * ratio and
    fed back as (delta * ratio) / ratio (src/RosMpc.cpp:85-90), which is not always delta in fp64."""
    N, B, steps, dtp, ratio = 40, 16, 5, 1.0 / 30.0, 17.3
    t, ws, pv = world_states(B, seed=33)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_track(t)
    wend, trace, nfail = s.closed_loop(ws, pv, steps, dt_plant=dtp, steering_ratio=ratio, trace=True,
                                       count_failures=True)
    wo, to = _oracle_closed_loop(t, ws, pv, N, steps, dtp, ratio=ratio)
    assert nfail == 0
    err = np.abs(trace - to).reshape(steps, B, 6).max(axis=2)
    assert err[0].max() <= 1e-7
    assert np.median(err[-1]) <= 1e-8 and (err[-1] <= 1e-5).mean() >= 0.9
    # the world path after a closed loop belongs to the pose the last front-end pass used, not the plant's
    xy = s.get_world_path(B)
    g = s.get(B)
    _, _, rot = s.get_frontend(B)
    pose = trace[-2] if steps > 1 else ws
    a = pose[:, 2] - rot
    wx = g["x"][:, :, 0] * np.cos(a)[:, None] - g["x"][:, :, 1] * np.sin(a)[:, None] + pose[:, 0:1]
    wy = g["x"][:, :, 0] * np.sin(a)[:, None] + g["x"][:, :, 1] * np.cos(a)[:, None] + pose[:, 1:2]
    assert np.abs(xy[:, :, 0] - wx).max() <= 1e-9 and np.abs(xy[:, :, 1] - wy).max() <= 1e-9
    s.close()


def test_config5_full_size_closed_loop_properties():
    """BASELINE config 5 at full size: 65,536 vehicles x 200 ticks, per-vehicle weights and reference
    speeds, warm start without shift, all on the device.  Size-independent properties: finite states,
    actuator limits, the fleet stays on the track and moves along it, almost every solve succeeds."""
    N, B, steps = 40, 65536, 200
    track = sc.test_track()
    ws, pv = sc.cars_on_track(B, track)
    rng = np.random.default_rng(5)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_track(track)
    s.set_weights(np.array(s.cfg.W[:])[None, :] * rng.uniform(0.5, 2.0, size=(B, 7)))
    wend, _, _ = s.closed_loop(ws, pv, steps)
    last = s.get_controls(B)[:, 5].astype(np.int64)
    assert np.isfinite(wend).all()
    assert (last % 16 == 0).mean() >= 0.999
    assert (np.abs(wend[:, 4]) <= 0.57 + 1e-9).all() and (np.abs(wend[:, 5]) <= 1.0 + 1e-9).all()
    sub = np.arange(0, B, 16)
    d = np.sqrt(((wend[sub, None, 0:2] - track[None, :, :]) ** 2).sum(-1)).min(axis=1)
    d0 = np.sqrt(((ws[sub, None, 0:2] - track[None, :, :]) ** 2).sum(-1)).min(axis=1)
    moved = np.sqrt(((wend[:, 0:2] - ws[:, 0:2]) ** 2).sum(-1))
    assert np.median(d) < 0.5 and np.median(d) <= np.median(d0) + 0.05
    assert np.median(moved) > 5.0
    # speeds head for the per-vehicle reference speed
    assert np.median(np.abs(wend[:, 3] - pv[:, 1])) < np.median(np.abs(ws[:, 3] - pv[:, 1]))
    s.close()


def test_sqp_iteration_limit_restarts_cold():
    """Status 2 (SQP iteration limit) is a failure like any other for the reference, which re-initialises
    on every status != ACADOS_SUCCESS (src/AcadosSolver.cpp:56-59): the next call starts from the cold
    guess, i.e. repeats the very same first iteration."""
    N, B = 40, 32
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0)
    s.set_params(p)
    s.solve(B, mode=capi.MODE_SQP, max_iter=1, tol=1e-13)
    g1 = s.get(B)
    assert (g1["status"] == 2).all() and (g1["sqp_iters"] == 1).all()
    s.solve(B, mode=capi.MODE_SQP, max_iter=1, tol=1e-13)
    g2 = s.get(B)
    assert (g2["status"] == 2).all()
    assert np.array_equal(g1["x"], g2["x"]) and np.array_equal(g1["u"], g2["u"])
    xo, uo, _, _ = orc.rti_batch(_ocfg(N), x0, p)       # one full step from the cold guess = one cold RTI
    assert_parity(g2, xo, uo)
    s.close()


def test_reset_clears_the_launch_order_history():
    """nmpc_reset forgets the iteration counts of reset instances (a cold solve's work is unrelated to
    the last warm one): after a full reset the history is flat again."""
    N, B = 40, 256
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0)
    s.set_params(p)
    s.solve(B)
    assert s.get(B)["ipm_iters"].min() > 0
    mask = np.zeros(B, dtype=np.uint8)
    mask[:100] = 1
    s.reset(mask)
    it = s.get(B)["ipm_iters"]
    assert (it[:100] == 0).all() and (it[100:] > 0).all()
    s.reset(None, B)
    assert (s.get(B)["ipm_iters"] == 0).all()
    s.close()


def test_weights_and_bounds_are_validated():
    N, B = 40, 4
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    l = capi.lib()
    W = np.tile(np.array(s.cfg.W[:]), (B, 1))
    W[2, 5] = 0.0                                    # singular input Hessian
    assert l.nmpc_set_weights(s.handle, W.ctypes.data_as(C.POINTER(C.c_double)), B) == capi.EINVAL
    assert b"W[2][5]" in l.nmpc_last_error()
    bnd = np.tile(np.array([-80, -0.33, -0.57, -1.0, 80, 0.33, 0.57, 1.0]), (B, 1))
    bnd[1, 6] = -0.57                                # lower == upper
    assert l.nmpc_set_bounds(s.handle, bnd.ctypes.data_as(C.POINTER(C.c_double)), B) == capi.EINVAL
    s.close()
    cfg = capi.default_config(N, 0.35)
    cfg.W[0] = -1.0
    h = C.c_void_p()
    assert l.nmpc_create(C.byref(cfg), 4, 0, C.byref(h)) == capi.EINVAL


def test_dpp_row_broadcast_canary():
    """v_fmac_f64 with a DPP row broadcast is what the vector sweeps are made of; its operand hazards are
    hand-managed (two wait states between a VALU write and a DPP read).  tools/micro/dpp_fmac_test.hip
    checks the instruction's semantics lane by lane on the device the tests run on."""
    exe = os.path.join(ROOT, "tests", "cpp", "dpp_canary")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "check: 0 bad lanes" in out.stdout


@pytest.mark.parametrize("N", [40, 30, 10, 63, 1])
def test_quad_kernel_matches_oracle(N, monkeypatch):
    """NMPC_KERNEL=quad: the four-slots-per-workgroup kernel (nmpc_quad_solver.h: one wave sweeps four
    instances, one per 16-lane DPP row, slots draw instances from a ticket queue) against the oracle:
    cold + two warm calls, batch sizes that leave slots empty, more instances than slots, and the failure
    statuses.  The default kernel is checked by test_gpu_parity.py."""
    monkeypatch.setenv("NMPC_KERNEL", "quad")
    ocfg = _ocfg(N, polish=0)       # (this kernel runs the interior-point iteration to the end: no active-set polish)
    for B in (1, 7, 1500):
        x0, p = sc.config2(B)
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
        s.set_x0(x0)
        s.set_params(p)
        xo = uo = None
        for it in range(3):
            s.solve(B)
            g = s.get(B)
            xo, uo, so, io = orc.rti_batch(ocfg, x0, p, xo, uo, cold=(it == 0))
            assert (g["status"] == 0).all() and (so == 0).all()
            if B >= 64:
                assert_parity(g, xo, uo, io, frac=0.99)
            else:
                assert (_err(g, xo, uo) <= 1e-9 * _scale(xo, uo)).all()
            assert (np.abs(g["ipm_iters"] - io) <= 1).all()
            ctrl = s.get_controls(B)
            assert np.array_equal(ctrl[:, 0:2], g["u"][:, 0, :])
            if N >= 2:
                assert np.array_equal(ctrl[:, 2:4], g["x"][:, 1, 4:6])
        s.close()
    # failure statuses: NaN measurement (1) for one instance only, interior-point iteration limit (4)
    B = 64
    x0, p = sc.config2(B)
    bad = x0.copy()
    bad[5, 3] = np.nan
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(bad)
    s.set_params(p)
    s.solve(B)
    g = s.get(B)
    assert g["status"][5] == 1 and (np.delete(g["status"], 5) == 0).all()
    s.set_x0(x0)
    s.solve(B)              # the failed one restarts cold, the others are warm
    assert (s.get(B)["status"] == 0).all()
    s.close()
    c = capi.default_config(N, 0.35, polish=0)
    c.ipm_max_iter = 3
    s = capi.BatchSolver(cfg=c, capacity=B)
    s.set_x0(x0)
    s.set_params(p)
    s.solve(B)
    g = s.get(B)
    xo, uo, so, io = orc.rti_batch(orc.config_like(c), x0, p)
    assert np.array_equal(g["status"], so) and np.array_equal(g["ipm_iters"], io)
    assert np.abs(g["x"] - xo).max() <= 1e-9 and np.abs(g["u"] - uo).max() <= 1e-9
    s.close()


def test_quad_kernel_full_batch_and_weights(monkeypatch):
    """The quad kernel on config 2 at full size with per-instance weights and bounds, against the
    default kernel on the same inputs (both within the bar of the oracle, hence of each other)."""
    N, B = 40, 4096
    x0, p = sc.config2(B)
    rng = np.random.default_rng(3)
    W = np.array(capi.default_config(N, 0.35).W[:])[None, :] * rng.uniform(0.5, 2.0, size=(B, 7))
    out = {}
    for kern in ("quad", "wave"):
        monkeypatch.setenv("NMPC_KERNEL", kern)
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B, polish=0)     # (the same algorithm on both: interior point to the end)
        s.set_x0(x0)
        s.set_params(p)
        s.set_weights(W)
        s.solve(B)
        s.solve(B)
        out[kern] = s.get(B)
        s.close()
    a, b = out["quad"], out["wave"]
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    sc_ = _scale(b["x"], b["u"])
    e = _err(a, b["x"], b["u"])
    assert e.max() <= TOL_BAR * sc_ and (e <= 1e-9 * sc_).mean() >= 0.99
    assert (np.abs(a["ipm_iters"] - b["ipm_iters"]) <= 1).all()


def test_kernel_choice_by_batch_and_horizon(monkeypatch):
    """Without NMPC_KERNEL and with the active-set polish off the library picks the four-slot kernel for
    32 <= N <= 43 and B >= 8192 and the one-instance kernel otherwise; with the polish on (the default) always
    the one-instance kernel, the only one that has it (nmpc_hip.hip: use_quad_kernel).  Seen from outside: the
    automatic choice is bit-identical to the forced kernel it should have picked, and both kernels agree to
    the bar."""
    def run(N, B, kern, polish=0):
        if kern is None:
            monkeypatch.delenv("NMPC_KERNEL", raising=False)
        else:
            monkeypatch.setenv("NMPC_KERNEL", kern)
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B, polish=polish)
        x0, p = sc.config2(B)
        s.set_x0(x0)
        s.set_params(p)
        s.solve(B)
        g = s.get(B)
        s.close()
        return g
    for N, B, expect in ((40, 8192, "quad"), (40, 4096, "wave"), (20, 8192, "wave"), (44, 8192, "wave")):
        auto, quad, wave = run(N, B, None), run(N, B, "quad"), run(N, B, "wave")
        same = quad if expect == "quad" else wave
        assert np.array_equal(auto["x"], same["x"]) and np.array_equal(auto["u"], same["u"]), (N, B)
        assert not np.array_equal(quad["x"], wave["x"])          # (the two kernels round differently)
        assert (_err(quad, wave["x"], wave["u"]) <= TOL_BAR * _scale(wave["x"], wave["u"])).all()
        assert (auto["status"] == 0).mean() > 0.999
    # the default configuration (polish on): the one-instance kernel whatever the batch, and the same minimisers
    auto, wave, quad = run(40, 8192, None, polish=1), run(40, 8192, "wave", polish=1), run(40, 8192, "quad")
    assert np.array_equal(auto["x"], wave["x"]) and np.array_equal(auto["u"], wave["u"])
    assert (_err(quad, wave["x"], wave["u"]) <= TOL_BAR * _scale(wave["x"], wave["u"])).all()
    assert auto["ipm_iters"].mean() < quad["ipm_iters"].mean() - 1.5


def test_horizon_sweep_inside_one_launch():
    """nmpc_set_horizons (north_star: horizon sweeps per launch): a batch whose instances have horizons
    20 / 30 / 40 / 64 / 100 gives, instance by instance and bit for bit, what a handle created with that
    horizon gives -- for a cold call and a warm one, in SQP-RTI and in SQP mode -- and so the oracle's result."""
    B, NMAX = 120, 100
    horizons = np.array([20, 30, 40, 64, 100], dtype=np.int32)[np.arange(B) % 5]
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=NMAX, dt=0.35, capacity=B)
    s.set_x0(x0)
    s.set_params(p)
    s.set_horizons(horizons)
    mixed = []
    for call in range(2):
        s.solve(B)
        mixed.append(s.get(B))
    for n in np.unique(horizons):
        sel = np.where(horizons == n)[0]
        one = capi.BatchSolver(N=int(n), dt=0.35, capacity=len(sel))
        one.set_x0(x0[sel])
        one.set_params(p[sel])
        xo = uo = None
        for call in range(2):
            one.solve(len(sel))
            g = one.get(len(sel))
            m = mixed[call]
            assert np.array_equal(m["x"][sel][:, :n + 1], g["x"]) and np.array_equal(m["u"][sel][:, :n], g["u"])
            assert np.array_equal(m["status"][sel], g["status"]) and np.array_equal(m["ipm_iters"][sel], g["ipm_iters"])
            assert np.array_equal(m["cost"][sel], g["cost"])
            assert (m["x"][sel][:, n + 1:] == 0).all()          # nodes beyond the instance's horizon are untouched
            if call == 0:
                xo, uo, so, io = orc.rti_batch(_ocfg(int(n)), x0[sel], p[sel])
                assert (so == 0).all() and (g["status"] == 0).all()
                assert (_err(g, xo, uo) <= 1e-9 * _scale(xo, uo)).all()
        one.close()
    # back to one horizon for all; SQP mode with mixed horizons
    s.set_horizons(None)
    s.reset(None, B)
    s.solve(B)
    full = s.get(B)
    assert (np.abs(full["x"][:, NMAX]).max(axis=1) > 0).all()
    s.set_horizons(horizons)
    s.reset(None, B)
    s.solve(B, mode=capi.MODE_SQP, max_iter=3, tol=1e-13)
    gs = s.get(B)
    for n in (20, 64):
        sel = np.where(horizons == n)[0]
        one = capi.BatchSolver(N=n, dt=0.35, capacity=len(sel))
        one.set_x0(x0[sel])
        one.set_params(p[sel])
        one.solve(len(sel), mode=capi.MODE_SQP, max_iter=3, tol=1e-13)
        g = one.get(len(sel))
        assert np.array_equal(gs["x"][sel][:, :n + 1], g["x"]) and np.array_equal(gs["sqp_iters"][sel], g["sqp_iters"])
        one.close()
    # argument checks
    l = capi.lib()
    bad = np.array([0, 5], dtype=np.int32)
    assert l.nmpc_set_horizons(s.handle, bad.ctypes.data_as(C.POINTER(C.c_int32)), 2) == capi.EINVAL
    bad = np.array([NMAX + 1], dtype=np.int32)
    assert l.nmpc_set_horizons(s.handle, bad.ctypes.data_as(C.POINTER(C.c_int32)), 1) == capi.EINVAL
    s.close()


@pytest.mark.parametrize("NMAX", [63, 127])
def test_horizon_sweep_edge_horizons(NMAX):
    """Per-instance horizons at the edges -- 1, 2, the last one-wave horizon and the first two-wave ones,
    the handle's own maximum -- inside a one-wave (NMAX = 63) and a two-wave (NMAX = 127) launch: three
    calls (cold, warm, warm), bit-identical to handles created with that horizon, SQP mode included."""
    hs = [h for h in (1, 2, 3, 62, 63, 64, 65, 126, 127) if h <= NMAX]
    B = 6 * len(hs)
    horizons = np.array(hs, dtype=np.int32)[np.arange(B) % len(hs)]
    x0, p = sc.config2(B, start=300)
    for mode_kw in ({}, dict(mode=capi.MODE_SQP, max_iter=4, tol=1e-12)):
        s = capi.BatchSolver(N=NMAX, dt=0.35, capacity=B)
        s.set_x0(x0)
        s.set_params(p)
        s.set_horizons(horizons)
        mixed = []
        for call in range(3):
            s.solve(B, **mode_kw)
            mixed.append(s.get(B))
        s.close()
        for n in hs:
            sel = np.where(horizons == n)[0]
            one = capi.BatchSolver(N=int(n), dt=0.35, capacity=len(sel))
            one.set_x0(x0[sel])
            one.set_params(p[sel])
            for call in range(3):
                one.solve(len(sel), **mode_kw)
                g = one.get(len(sel))
                m = mixed[call]
                assert np.array_equal(m["x"][sel][:, :n + 1], g["x"]), (n, call, mode_kw)
                assert np.array_equal(m["u"][sel][:, :n], g["u"]), (n, call, mode_kw)
                assert np.array_equal(m["status"][sel], g["status"])
                assert np.array_equal(m["ipm_iters"][sel], g["ipm_iters"])
                assert np.array_equal(m["sqp_iters"][sel], g["sqp_iters"])
            one.close()


def test_gather_controls_over_rccl_single_rank():
    """nmpc_gather_controls through the C ABI with a real RCCL communicator.  The test box has one GPU,
    so the communicator has one rank (ncclCommInitAll): the gathered block must equal this rank's control
    records, for the handle's own buffer and for a caller's buffer (nmpc_set_controls_device); the
    two-rank logic of the collective path is covered on CPU by tests/test_dist_gloo.py."""
    import torch
    rccl = C.CDLL("librccl.so.1")
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        N, B = 40, 512
        x0, p = sc.config2(B)
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
        s.set_x0(x0)
        s.set_params(p)
        out = torch.full((B * 6,), -1.0, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        l = capi.lib()
        s.solve(B)
        assert l.nmpc_gather_controls(s.handle, comm, B, C.c_void_p(out.data_ptr())) == capi.OK, l.nmpc_last_error()
        s.sync()
        assert np.array_equal(out.cpu().numpy().reshape(B, 6), s.get_controls(B))
        buf = torch.zeros(B * 6, dtype=torch.float64, device="cuda:0")
        s.set_controls_device(buf.data_ptr())
        s.solve(B)                                       # warm call into the caller's buffer
        assert l.nmpc_gather_controls(s.handle, comm, 100, C.c_void_p(out.data_ptr())) == capi.OK
        s.sync()
        got = out.cpu().numpy().reshape(B, 6)
        assert np.array_equal(got[:100], buf.cpu().numpy().reshape(B, 6)[:100])
        assert l.nmpc_gather_controls(s.handle, None, B, C.c_void_p(out.data_ptr())) == capi.EINVAL
        assert l.nmpc_gather_controls(s.handle, comm, B + 1, C.c_void_p(out.data_ptr())) == capi.EINVAL
        s.close()
    finally:
        rccl.ncclCommDestroy(comm)


def test_stalled_interior_point_iterations_fail_early():
    """A QP on which the interior-point iteration makes no headway (infeasibility scale still above
    ipm_stall_lin after ipm_stall_iter iterations) is reported as QP failure (status 4) at that point, on
    the kernel as on the oracle, instead of holding a launch up until ipm_max_iter.  Found on a warm
    sequence of config 2 (a few instances per 4096 and call); with the rule switched off the same
    instances run to the limit and fail there."""
    N, B = 40, 4096
    host = [sc.config2(B, start=s * B) for s in range(2)]
    x1 = host[0][0] + 0.03 * (host[1][0] - host[0][0])
    p1 = host[0][1] + 0.03 * (host[1][1] - host[0][1])
    out = {}
    for stall in (20, 0):
        c = capi.default_config(N, 0.35)
        c.ipm_stall_iter = stall
        s = capi.BatchSolver(cfg=c, capacity=B)
        s.set_x0(host[0][0]); s.set_params(host[0][1])
        s.solve(B)
        s.set_x0(x1); s.set_params(p1)
        s.solve(B)                                  # warm call on moved inputs
        g = s.get(B)
        oc = orc.config_like(c)
        hint = np.zeros(B)
        xo, uo, _, _ = orc.rti_batch(oc, host[0][0], host[0][1], hint=hint)
        xo, uo, so, io = orc.rti_batch(oc, x1, p1, xo, uo, cold=False, hint=hint)
        assert np.array_equal(g["status"], so)
        assert (np.abs(g["ipm_iters"] - io) <= 1).all()
        ok = so == 0
        assert_parity(g, xo, uo, io, ok=ok)
        out[stall] = (g["status"].copy(), g["ipm_iters"].copy())
        s.close()
    st_on, it_on = out[20]
    st_off, it_off = out[0]
    failed = st_off == 4
    # (reported counts include the active-set solves of attempts that were not accepted: a handful at most)
    assert failed.sum() >= 1 and (it_off[failed] >= 50).all() and (it_off[failed] <= 56).all()
    assert np.array_equal(st_on == 4, failed)                  # the same instances fail, ...
    assert (it_on[failed] >= 20).all() and (it_on[failed] <= 26).all()   # ... thirty iterations earlier
    assert np.array_equal(it_on[~failed], it_off[~failed])     # nothing else changes


def test_every_horizon_matches_the_oracle():
    """All horizons 1..127 -- across the one-wave / two-wave boundary and every switch between the 54- and
    the 48-double stage layout (nmpcw::compact_layout: 49|50, 55|56, 58|59, 63|64, 90|91, 102|103,
    121|122) -- cold call and one warm call (re-seeded from the oracle's iterate, as in
    test_cold_then_warm_rti_matches_oracle), four instances each, against the oracle."""
    B = 4
    x0, p = sc.config2(B, start=900)
    x1, p1 = sc.config2(B, start=904)
    xw, pw = x0 + 0.05 * (x1 - x0), p + 0.05 * (p1 - p)
    worst = 0.0
    for N in range(1, 128):
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
        s.set_x0(x0)
        s.set_params(p)
        s.solve(B)
        g = s.get(B)
        ocfg = _ocfg(N)
        xo, uo, so, io = orc.rti_batch(ocfg, x0, p)
        assert np.array_equal(g["status"], so), N
        ok = so == 0
        e = (_err(g, xo, uo) / _scale(xo, uo))[ok]
        assert (e <= TOL_BAR).all(), (N, e.max())
        assert (np.abs(g["ipm_iters"] - io) <= 1).all(), N
        worst = max(worst, e.max() if e.size else 0.0)
        # warm call on a moved measurement, both sides from the oracle's iterate
        s.set_iterate(xo, uo)
        s.set_x0(xw)
        s.set_params(pw)
        s.solve(B)
        g = s.get(B)
        xo2, uo2, so2, _ = orc.rti_batch(ocfg, xw, pw, xo.copy(), uo.copy(), cold=False)
        assert np.array_equal(g["status"], so2), N
        ok = so2 == 0
        e = (_err(g, xo2, uo2) / _scale(xo2, uo2))[ok]
        assert (e <= TOL_BAR).all(), (N, "warm", e.max())
        s.close()
    assert worst <= TOL_BAR


@pytest.mark.parametrize("N", [50, 58, 59, 63, 64, 90, 91, 102, 103, 121, 122])
def test_sqp_mode_at_the_layout_boundaries(N):
    """SQP mode (multipliers and costates wait in the LDS image between QPs) on either side of every
    layout switch: same status, same number of QPs and the same iterate as the oracle after three QPs."""
    B = 6
    x0, p = sc.config2(B, start=40)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0)
    s.set_params(p)
    s.solve(B, mode=capi.MODE_SQP, max_iter=3, tol=1e-13)
    g = s.get(B)
    s.close()
    ocfg = _ocfg(N)
    for i in range(B):
        xo, uo = orc.init_guess(ocfg, x0[i])
        st = orc.sqp(ocfg, x0[i], p[i], xo, uo, 3, 1e-13)
        assert g["status"][i] == st.status and g["sqp_iters"][i] == st.sqp_iters, (N, i)
        if st.status in (0, 2):
            sc_ = max(1.0, np.abs(xo).max(), np.abs(uo).max())
            assert np.abs(g["x"][i] - xo).max() <= TOL_BAR * sc_ and np.abs(g["u"][i] - uo).max() <= TOL_BAR * sc_, (N, i)


@pytest.mark.parametrize("kern", ["quad", "wave"])
def test_ill_conditioned_stage_costs_both_kernels(kern, monkeypatch):
    """GPU counterpart of tests/test_emu_wave.py::test_emulated_kernels_on_an_ill_conditioned_stage_cost,
    on a whole neighbourhood of the stress batch (warm call, 1 % of the neighbour mixed into every
    instance's inputs: path curvature coefficients up to the hundreds): every instance the oracle solves
    stays within the bar, with the oracle's iteration count."""
    monkeypatch.setenv("NMPC_KERNEL", kern)
    B = 512
    xa, pa = sc.config2(B + 1, start=20_000_000 + 16767 - 200)
    x1 = xa[1:] + 0.01 * (xa[:-1] - xa[1:])
    p1 = pa[1:] + 0.01 * (pa[:-1] - pa[1:])
    for N in (40, 63):
        ocfg = _ocfg(N, polish=0 if kern == "quad" else 1)
        xo, uo, so, _ = orc.rti_batch(ocfg, xa[1:], pa[1:])
        xo2, uo2, so2, io2 = orc.rti_batch(ocfg, x1, p1, xo.copy(), uo.copy(), cold=False)
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
        s.set_iterate(xo, uo)
        s.set_x0(x1)
        s.set_params(p1)
        s.solve(B)
        g = s.get(B)
        s.close()
        ok = so2 == 0
        assert ok.mean() > 0.99 and np.array_equal(g["status"][ok], so2[ok])
        e = (_err(g, xo2, uo2) / _scale(xo2, uo2))[ok]
        assert e.max() <= TOL_BAR, (N, kern, e.max())
        assert (np.abs(g["ipm_iters"] - io2)[ok] <= 1).all()
