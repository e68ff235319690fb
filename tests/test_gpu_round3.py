"""GPU tests added in round 3 (each through the C ABI, against the oracle): the active-set polish of the QP
(iteration counts, exact vertices, warm-start hints), the remaining gaps the round-2 review listed (front-end
4-point / wrap windows, pinned stress outliers, config 4 at full size, free-running chains at the wave-count
switch, the second actuation branch), trajectories through pinned host memory, and the fixes to
nmpc_solve_static / nmpc_set_horizons."""
import ctypes as C

import numpy as np
import pytest

from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
from test_frontend_emu import world_states
from test_gpu_parity import TOL_BAR, _err, _ocfg, _scale

pytestmark = pytest.mark.gpu


def test_polish_cuts_iterations_and_lands_on_the_vertex():
    """Config 2 at full size, cold: with the active-set polish the factorisation count drops by a quarter
    (mean 7.4, 99.8 % within 12 against 9.9 / 20 of the interior-point iteration alone), every instance agrees
    with the oracle to 1e-9 (no interior-point tail left), quantities the solution puts on a bound sit there
    to round-off, and the polish-off configuration reproduces the interior-point result to its accuracy."""
    N, B = 40, 4096
    x0, p = sc.config2(B)
    out = {}
    for polish in (1, 0):
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B, polish=polish)
        s.set_x0(x0); s.set_params(p)
        s.solve(B)
        out[polish] = s.get(B)
        s.close()
    g, g0 = out[1], out[0]
    assert (g["status"] == 0).all() and (g0["status"] == 0).all()
    # (two or three instances in 4096 need 13 .. 15: their interior-point iteration alone takes 12 .. 13)
    it = g["ipm_iters"]
    assert it.mean() <= 7.6 and (it <= 12).mean() >= 0.998 and it.max() <= 16, (it.mean(), it.max())
    assert g0["ipm_iters"].mean() >= 9.5 and g0["ipm_iters"].max() >= 18
    sub = slice(0, 512)
    xo, uo, so, io = orc.rti_batch(_ocfg(N), x0[sub], p[sub])
    e = _err(dict(x=g["x"][sub], u=g["u"][sub]), xo, uo) / _scale(xo, uo)
    assert e.max() <= 1e-9, e.max()                           # 100 %, not 99 %
    assert np.array_equal(g["ipm_iters"][sub], io)
    # the vertex: u1 (throttle rate, +-0.33) is on a bound somewhere in almost every instance -- exactly
    u1 = g["u"][:, :, 1]
    near = np.abs(np.abs(u1) - 0.33) < 1e-6
    assert near.any(axis=1).mean() > 0.9
    assert np.abs(np.abs(u1[near]) - 0.33).max() <= 1e-15
    # interior point alone: the same minimisers to ITS accuracy
    e0 = _err(g0, g["x"], g["u"]) / _scale(g["x"], g["u"])
    assert e0.max() <= TOL_BAR and np.median(e0) <= 1e-11


def test_warm_hint_travels_with_the_iterate():
    """A closed-loop style sequence at steady state: after a few ticks most instances are solved by ONE
    active-set solve on the set their iterate sits on (the hint lives in the iterate buffer, so it survives
    nmpc_get / a second handle fed by nmpc_set_iterate loses it and is merely slower, not different)."""
    N, B = 40, 256
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    ocfg = _ocfg(N)
    hint = np.zeros(B)
    xo = uo = None
    its = []
    ticks = 8
    for tick in range(ticks):
        xk = x0 + 0.002 * tick * np.array([0, 0, 0.1, 1.0, 0, 0])      # a slowly moving measurement
        if tick == ticks - 1:
            x_before, u_before = xo.copy(), uo.copy()
        s.set_x0(xk); s.set_params(p)
        s.solve(B)
        g = s.get(B)
        xo, uo, so, io = orc.rti_batch(ocfg, xk, p, xo, uo, cold=(tick == 0), hint=hint)
        ok = (so == 0) & (g["status"] == 0)
        assert ok.mean() > 0.99
        e = (_err(g, xo, uo) / _scale(xo, uo))[ok]
        assert e.max() <= TOL_BAR and np.median(e) <= 1e-10
        assert (g["ipm_iters"][ok] == io[ok]).mean() >= 0.95
        its.append(g["ipm_iters"].mean())
    assert its[-1] <= its[0] - 3.0 and its[0] >= 6.0, its       # (7.5 on the cold call, under 4 by the eighth)
    s.close()
    # the same last tick on a handle that only got the iterate (no hint): the same result with more work
    s2 = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s2.set_iterate(x_before, u_before)
    s2.set_x0(xk); s2.set_params(p)
    s2.solve(B)
    g2 = s2.get(B)
    s2.close()
    ok = (g["status"] == 0) & (g2["status"] == 0)
    assert (_err(dict(x=g2["x"][ok], u=g2["u"][ok]), g["x"][ok], g["u"][ok]) <= 1e-7 * _scale(g["x"], g["u"])).all()
    assert g2["ipm_iters"].mean() > g["ipm_iters"].mean() + 1.0


@pytest.mark.parametrize("N", [63, 64])
def test_free_running_rti_chain_at_the_wave_count_switch(N):
    """1 cold + 4 warm SQP-RTI calls at N = 63 (one wave, 48-double stage) and N = 64 (two waves) without
    re-seeding the kernel from the oracle's iterate: every instance stays within the bar on every call."""
    B = 64
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0); s.set_params(p)
    ocfg = _ocfg(N)
    hint = np.zeros(B)
    xo = uo = None
    for call in range(5):
        s.solve(B)
        g = s.get(B)
        xo, uo, so, io = orc.rti_batch(ocfg, x0, p, xo, uo, cold=(call == 0), hint=hint)
        assert (so == 0).all() and (g["status"] == 0).all()
        e = _err(g, xo, uo) / _scale(xo, uo)
        assert e.max() <= TOL_BAR, (call, e.max())
        assert np.median(e) <= 1e-10, (call, np.median(e))
    s.close()


def test_frontend_four_point_and_wrap_windows():
    """The window rules of getTrackSection (src/PathTrackingSys.cpp:128-134): cars whose nearest point is so
    close to the end of the track that fewer than four points are left get the window [start, start + 4),
    and when that runs past the end, the wrap window [0, 4).  A cubic through four points is exact in every
    rotated frame, so WHICH rotation wins is decided by round-off (not compared); what is compared: the fit
    of the chosen frame interpolates the window's four points, i.e. the window is the reference's."""
    t = orc.test_track()
    n = t.shape[0]
    ws, want = [], []
    for idx, off in ((n - 1, 0.02), (n - 2, 0.03), (n - 3, -0.02), (n - 4, 0.04), (n - 5, 0.01)):
        w = np.zeros(6)
        w[0:2] = t[idx] + np.array([off, -off])
        d = t[idx] - t[idx - 1]
        w[2] = np.arctan2(d[1], d[0]) + 0.05
        w[3] = 4.0
        st, en = orc.track_section(t, w[0], w[1])
        if en - st == 4:
            ws.append(w); want.append((st, en))
    assert len(ws) >= 4 and any(st == 0 for st, _ in want) and any(st > 0 for st, _ in want)
    ws = np.array(ws)
    pv = np.tile([0.0, 5.0], (len(ws), 1))
    s = capi.BatchSolver(N=40, dt=0.35, capacity=len(ws))
    s.set_track(t)
    s.frontend(ws, pv)
    x0s, par, rot = s.get_frontend()
    s.close()
    rots = [-np.pi / 2, -np.pi / 2 + np.pi / 6, -np.pi / 2 + 2 * (np.pi / 6), -np.pi / 2 + 3 * (np.pi / 6)]
    for i, (st, en) in enumerate(want):
        assert min(abs(rot[i] - r) for r in rots) < 1e-12
        a = rot[i] - ws[i, 2]
        dx, dy = t[st:en, 0] - ws[i, 0], t[st:en, 1] - ws[i, 1]
        lx = dx * np.cos(a) - dy * np.sin(a)
        ly = dx * np.sin(a) + dy * np.cos(a)
        fit = par[i, 0] + par[i, 1] * lx + par[i, 2] * lx ** 2 + par[i, 3] * lx ** 3
        assert np.abs(fit - ly).max() <= 1e-9 * max(1.0, np.abs(ly).max()), (i, st, en)
        assert np.array_equal(x0s[i], [0.0, 0.0, rot[i], ws[i, 3], ws[i, 4], ws[i, 5]])


def test_config4_full_size_status_and_iterations_on_a_sample():
    """Config 4 at full size (4096 x N = 100, full SQP, <= 20 iterations, tol 1e-6): status and SQP iteration
    count equal to the oracle's on a strided sample of 128 instances (full-step SQP without globalisation
    converges for about 4 in 10 of these hard starts on both sides), iterates of the converged ones within
    the bar."""
    N, B = 100, 4096
    x0, p = sc.config4(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0); s.set_params(p)
    s.solve(B, mode=capi.MODE_SQP, max_iter=20, tol=1e-6)
    g = s.get(B)
    s.close()
    ocfg = _ocfg(N)
    sel = np.arange(0, B, 32)
    same = conv = 0
    for i in sel:
        xo, uo = orc.init_guess(ocfg, x0[i])
        st = orc.sqp(ocfg, x0[i], p[i], xo, uo, 20, 1e-6)
        same += int(g["status"][i] == st.status and abs(int(g["sqp_iters"][i]) - st.sqp_iters) <= 1)
        if st.status == 0 and g["status"][i] == 0:
            conv += 1
            sc_ = max(1.0, np.abs(xo).max())
            assert np.abs(g["x"][i] - xo).max() <= 1e-5 * sc_ and np.abs(g["u"][i] - uo).max() <= 1e-5 * sc_, i
    assert same >= len(sel) - 2, (same, len(sel))
    assert conv >= 0.25 * len(sel)
    assert set(np.unique(g["status"])) <= {0, 2, 4}


# index within config2(start=40_000_000) of the instance on which this round's build with the polish OFF is
# furthest from the oracle (1.5e-8; tools/find_r2_outlier.py, profiles/r3/stress_n127_polish_off_vs_on.txt -- the
# 4.8e-4 of round 2's build is not reproduced by this one); with the polish on the whole 20,000-instance batch is
# within 1.4e-13
OUTLIER_N127 = 10267


def test_pinned_stress_outliers_are_gone():
    """The instances the round-2 stress runs left outside 1e-6 (profiles/r2/stress_parity.txt:97: N = 127 cold,
    seed 40000000, a degenerate slack/multiplier pair -- 4.8e-4 between kernel and oracle; stress_sqp.txt:
    three SQP-mode instances at 2e-5 .. 2e-6) were interior-point termination noise on weakly active bounds.
    With the QP finished on its vertex both sides return the same point: whole neighbourhoods of those
    batches within 1e-7, the N = 127 instance itself within 1e-9."""
    N, start = 127, 40_000_000
    i0 = OUTLIER_N127
    lo = max(0, i0 - 32)
    x0, p = sc.config2(64, start=start + lo)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=64)
    s.set_x0(x0); s.set_params(p)
    s.solve(64)
    g = s.get(64)
    s.close()
    xo, uo, so, io = orc.rti_batch(_ocfg(N), x0, p)
    assert (so == 0).all() and (g["status"] == 0).all()
    e = _err(g, xo, uo) / _scale(xo, uo)
    assert e.max() <= 1e-10 and e[i0 - lo] <= 1e-10, (e.max(), e[i0 - lo])
    # SQP mode, four chained full steps, on the batch whose worst instance was 2e-5 off (N = 63, config 4)
    N, B = 63, 128
    x0, p = sc.config4(B, start=70_000_000)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0); s.set_params(p)
    s.solve(B, mode=capi.MODE_SQP, max_iter=4, tol=1e-12)
    g = s.get(B)
    s.close()
    ocfg = _ocfg(N)
    worst = 0.0
    for i in range(B):
        xo, uo = orc.init_guess(ocfg, x0[i])
        st = orc.sqp(ocfg, x0[i], p[i], xo, uo, 4, 1e-12)
        assert g["status"][i] == st.status and g["sqp_iters"][i] == st.sqp_iters
        worst = max(worst, max(np.abs(g["x"][i] - xo).max(), np.abs(g["u"][i] - uo).max()) / max(1.0, np.abs(xo).max()))
    assert worst <= 1e-6, worst




def test_closed_loop_with_measured_steering_matches_the_oracle_loop():
    """use_actual_steering_topic = true (src/RosMpc.cpp:25,88-90,150): the state's steering angle is the
    measured steering-wheel angle / steering_ratio, not the command fed back; the car moves with the angle it
    has.  Against the oracle loop over a few ticks, with a ratio of 17.3 and measurements that lag the
    commands; the published commands are returned."""
    N, B, steps, dtp, ratio = 40, 16, 5, 1.0 / 30.0, 17.3
    t, ws, pv = world_states(B, seed=7)
    rng = np.random.default_rng(5)
    meas = ratio * (ws[None, :, 4] + 0.02 * rng.standard_normal((steps, B)))       # steering-wheel angles per tick
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_track(t)
    wend, trace, nfail, cmd = s.closed_loop(ws, pv, steps, dt_plant=dtp, steering_ratio=ratio, trace=True,
                                            count_failures=True, steer_meas=meas)
    s.close()
    assert nfail == 0
    ocfg = _ocfg(N)
    pcfg = orc.default_config(N, dtp)
    for i in range(B):
        w = ws[i].copy()
        xt = ut = None
        hint = np.zeros(1)
        for k in range(steps):
            w[4] = meas[k, i] / ratio
            rot, c = orc.calc_coeffs(t, w)
            xs = np.array([0.0, 0.0, rot, w[3], w[4], w[5]])
            pp = np.array([c[0], c[1], c[2], c[3], pv[i, 0], pv[i, 1]])
            if xt is None:
                xt, ut = orc.init_guess(ocfg, xs)
            assert orc.rti(ocfg, xs, pp, xt, ut, hint=hint).status == 0
            assert abs(cmd[k, i, 0] - xt[1, 4] * ratio) <= 1e-6 and abs(cmd[k, i, 1] - xt[1, 5]) <= 1e-6, (i, k)
            x = w.copy(); x[5] = xt[1, 5]
            phi, _, _ = orc.rk4_sens(pcfg, x, np.zeros(2), np.array([0, 0, 0, 0, pv[i, 0], 0.0]))
            w[:4] = phi[:4]; w[5] = xt[1, 5]
            assert np.abs(trace[k, i] - w).max() <= 1e-6 * max(1.0, np.abs(w).max()), (i, k)
    assert np.array_equal(wend, trace[-1])


def test_trajectories_through_pinned_host_memory():
    """nmpc_get_view: x / u split on the device, one DMA each into pinned host memory of the handle -- the same
    numbers nmpc_get copies out, for a full and for a partial batch, and again after another solve."""
    N, B = 40, 300
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0); s.set_params(p)
    s2 = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    for call, Bn in enumerate((B, 77, B)):
        s.solve(Bn)
        g = s.get(Bn)
        xv, uv = s.get_view(Bn)
        assert xv.shape == (Bn, N + 1, 6) and uv.shape == (Bn, N, 2)
        assert np.array_equal(xv, g["x"]) and np.array_equal(uv, g["u"])
        assert np.array_equal(s.get_controls(Bn)[:, 0:2], uv[:, 0, :])
        # the device-side split against the host-side packing of nmpc_set_iterate (independent code): a round trip
        rng = np.random.default_rng(call)
        xr, ur = rng.standard_normal((Bn, N + 1, 6)), rng.standard_normal((Bn, N, 2))
        s2.set_iterate(xr, ur)
        xv2, uv2 = s2.get_view(Bn)
        assert np.array_equal(xv2, xr) and np.array_equal(uv2, ur)
    s.close(); s2.close()


def test_static_entry_restarts_failed_instances_cold():
    """nmpc_solve_static on a warm call reads the caller's previous statuses: an instance that failed the call
    before (here: NaN measurement -> status 1) starts cold, as in nmpc_solve and as the reference re-initialises
    after every failure (src/AcadosSolver.cpp:56-59) -- bit-identical to nmpc_solve on both calls."""
    import torch
    N, B = 40, 512
    dev = torch.device("cuda:0")
    x0, p = sc.config2(B)
    bad = x0.copy()
    failing = np.arange(5, B, 37)
    bad[failing, 3] = np.nan
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    ref = []
    for xa in (bad, x0):
        s.set_x0(xa); s.set_params(p)
        s.solve(B)
        ref.append((s.get(B), s.get_controls(B)))
    cfg = s.cfg
    s.close()
    assert (ref[0][0]["status"][failing] == 1).all() and (ref[1][0]["status"] == 0).all()
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev)
    ptr = lambda v: C.c_void_p(v.data_ptr())
    traj = torch.zeros(B * (N + 1) * 8, dtype=torch.float64, device=dev)
    ctrl = torch.zeros(B * 6, dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    dp = t(p)
    for call, xa in enumerate((bad, x0)):
        dx0 = t(xa)
        torch.cuda.synchronize()
        rc = capi.lib().nmpc_solve_static(C.byref(cfg), 0, None, B, 1 if call == 0 else 0, ptr(dx0), ptr(dp), None, None,
                                          ptr(traj), ptr(ctrl), ptr(status), ptr(iters))
        assert rc == capi.OK, capi.lib().nmpc_last_error()
        torch.cuda.synchronize()
        g, rctrl = ref[call]
        tr = traj.cpu().numpy().reshape(B, N + 1, 8)
        ok = g["status"] == 0
        assert np.array_equal(status.cpu().numpy(), g["status"])
        assert np.array_equal(tr[ok][:, :, :6], g["x"][ok]) and np.array_equal(tr[ok][:, :N, 6:8], g["u"][ok])
        assert np.array_equal(ctrl.cpu().numpy().reshape(B, 6)[ok], rctrl[ok])
        assert np.array_equal(iters.cpu().numpy(), g["ipm_iters"])
    # the failed instances of call 1 were solved COLD in call 2: the oracle's cold solve, not a warm one
    xo, uo, so, _ = orc.rti_batch(_ocfg(N), x0[failing], p[failing])
    tr = traj.cpu().numpy().reshape(B, N + 1, 8)[failing]
    assert np.abs(tr[:, :, :6] - xo).max() <= 1e-9 * _scale(xo, uo)


def test_set_horizons_null_restores_the_handles_horizon_for_everyone():
    """nmpc_set_horizons(h, NULL) makes every instance use cfg.N again -- also those a later call does not
    name: set(N1 for 100), set(NULL), set(N2 for 50) leaves instances 50..99 at cfg.N, not at N1."""
    NMAX, B = 40, 100
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=NMAX, dt=0.35, capacity=B)
    s.set_x0(x0); s.set_params(p)
    s.set_horizons(np.full(B, 10, dtype=np.int32))
    s.solve(B)
    s.set_horizons(None)
    s.set_horizons(np.full(50, 20, dtype=np.int32))
    s.reset(None, B)
    s.solve(B)
    g = s.get(B)
    s.close()
    for n, sel in ((20, slice(0, 50)), (NMAX, slice(50, 100))):
        one = capi.BatchSolver(N=n, dt=0.35, capacity=50)
        one.set_x0(x0[sel]); one.set_params(p[sel])
        one.solve(50)
        r = one.get(50)
        one.close()
        assert np.array_equal(g["x"][sel][:, :n + 1], r["x"]) and np.array_equal(g["u"][sel][:, :n], r["u"]), n


def test_gather_pipeline_cuda_sync_single_rank_finish_blocks_the_host():
    """ControlGatherPipeline with CudaSync (the bench's N > 1 path) on one rank over RCCL: every step's
    gathered block equals the records of that step's solve, and finish() has really waited on the host --
    all collectives' events are complete when it returns (an Event.wait() would only have queued a stream
    wait), so the gathered block may be handed to code outside torch's streams."""
    import os
    import torch
    import torch.distributed as dist
    from mpc_local_planner_amd import sharding
    N, B, steps = 40, 512, 5
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29577"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
        solve_stream = torch.cuda.ExternalStream(int(s.stream()), device=dev)
        records = [torch.zeros(B * 6, dtype=torch.float64, device=dev) for _ in range(2)]
        gathered = [torch.full((B * 6,), -1.0, dtype=torch.float64, device=dev) for _ in range(2)]
        pipe = sharding.ControlGatherPipeline(records, gathered, sharding.CudaSync(solve_stream, dev))
        want = []
        seen = []
        for k in range(steps):
            x0, p = sc.config2(B, start=k * B)
            s.set_x0(x0); s.set_params(p)
            s.reset(None, B)
            i = pipe.begin_step()
            s.set_controls_device(records[i].data_ptr())
            s.solve(B)
            pipe.end_step()
            seen.append(pipe.tokens[i])
            if k >= steps - 2:
                # (the handle's own record buffer is not written while a caller's buffer is set: expected
                # values come from a second, plain solve of the same inputs below)
                want.append((i, x0, p))
        pipe.finish()
        assert all(t.query() for t in seen), "finish() returned before the collectives had completed"
        assert pipe.tokens == [None, None]
        i_last, g_last = pipe.last()
        assert i_last == want[-1][0]
        got = {i: gathered[i].clone() for i in (0, 1)}
        s.set_controls_device(None)
        for i, x0, p in want:
            s.set_x0(x0); s.set_params(p)
            s.reset(None, B)
            s.solve(B)
            ref = s.get_controls(B)
            assert np.array_equal(got[i].cpu().numpy().reshape(B, 6), ref)
            assert np.array_equal(records[i].cpu().numpy().reshape(B, 6), ref)
        s.close()
    finally:
        dist.destroy_process_group()


def test_closed_loop_forty_ticks_free_running_against_the_oracle_loop():
    """The reference's operating mode over many ticks (src/main.cpp:14-26, src/PathTrackingSys.cpp:58-61: iterate
    kept without shift, one QP per tick), free-running on both sides, the oracle carrying its own warm-start
    hints: with every QP finished on its vertex round-off is no longer amplified from tick to tick, so the
    world states still agree after 40 ticks, and the last tick takes the same number of factorisations -- warm
    attempts, corrections and cold paths included."""
    from test_gpu_frontend_closed_loop import world_states as ws_gen
    N, B, steps, dtp = 40, 32, 40, 1.0 / 30.0
    t, ws, pv = ws_gen(B, seed=11)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_track(t)
    wend, trace, nfail = s.closed_loop(ws, pv, steps, dt_plant=dtp, trace=True, count_failures=True)
    g = s.get(B)
    s.close()
    assert nfail == 0
    pcfg, cfg = orc.default_config(N, dtp), _ocfg(N)
    to = np.zeros((steps, B, 6))
    its = np.zeros(B, dtype=int)
    for i in range(B):
        w = ws[i].copy()
        xt = ut = None
        hint = np.zeros(1)
        for k in range(steps):
            rot, c = orc.calc_coeffs(t, w)
            xs = np.array([0.0, 0.0, rot, w[3], w[4], w[5]])
            p = np.array([c[0], c[1], c[2], c[3], pv[i, 0], pv[i, 1]])
            if xt is None:
                xt, ut = orc.init_guess(cfg, xs)
                hint[:] = 0
            st = orc.rti(cfg, xs, p, xt, ut, hint=hint)
            assert st.status == 0
            its[i] = st.ipm_iters
            x = w.copy(); x[4] = xt[1, 4]; x[5] = xt[1, 5]
            phi, _, _ = orc.rk4_sens(pcfg, x, np.zeros(2), np.array([0, 0, 0, 0, pv[i, 0], 0.0]))
            w[:4] = phi[:4]; w[4] = xt[1, 4]; w[5] = xt[1, 5]
            to[k, i] = w
    err = np.abs(trace - to).max(axis=2)                    # [steps, B]
    assert err[0].max() <= 1e-12 and err.max() <= 1e-8, (err[0].max(), err.max())
    assert np.median(err[-1]) <= 1e-12
    assert (g["ipm_iters"] == its).mean() >= 0.9, (g["ipm_iters"], its)
