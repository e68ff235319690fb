"""ctypes binding of the CPU lane emulator (tests/emu/): runs the kernel body
mpc_local_planner_amd/csrc/nmpc_wave_solver.h on 64 emulated lanes.  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "emu", "libnmpc_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        csrc = os.path.join(_HERE, "..", "mpc_local_planner_amd", "csrc")
        src = [os.path.join(_HERE, "emu", "emu_wave.cpp"), os.path.join(_HERE, "emu", "emu_frontend.cpp"),
               os.path.join(csrc, "nmpc_wave_solver.h"), os.path.join(csrc, "nmpc_quad_solver.h"),
               os.path.join(csrc, "nmpc_frontend.h")]
        if not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
            subprocess.check_call(["make", "-C", os.path.join(_HERE, "emu"), "-s", "-B"])
        _lib = C.CDLL(_LIB)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        d, i = C.c_double, C.c_int
        _lib.emu_solve_batch.argtypes = [i, d, d, i, d, d, d, d, d, d, i, d, i,
                                         dp, dp, dp, dp, dp, dp, ip, ip, ip, dp, i, i, d,
                                         i, d, i, d, i, i, d]
        _lib.emu_solve_batch_quad.argtypes = [i, d, d, i, d, d, d, d, d, d, i,
                                              dp, dp, dp, dp, dp, dp, ip, ip, ip, dp, i, i, i, d]
        _lib.emu_frontend.argtypes = [dp, i, dp, dp, i, dp, dp, dp]
        _lib.emu_plant_step.argtypes = [dp, d, d, d, d, d, d]
        _lib.emu_unrotate.argtypes = [dp, d, d, d, dp]
    return _lib


def frontend(track, ws, pv):
    """nmpc_frontend.h compiled for the CPU: solver-frame state, parameters, rotation."""
    track = np.ascontiguousarray(track, dtype=np.float64)
    ws = np.ascontiguousarray(ws, dtype=np.float64).reshape(-1, 6)
    pv = np.ascontiguousarray(pv, dtype=np.float64).reshape(-1, 2)
    B = ws.shape[0]
    x0s, par, rot = np.zeros((B, 6)), np.zeros((B, 6)), np.zeros(B)
    dd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    lib().emu_frontend(dd(track), track.shape[0], dd(ws), dd(pv), B, dd(x0s), dd(par), dd(rot))
    return x0s, par, rot


def plant_step(ws, delta_cmd, thr_cmd, pitch, dtp, wheelbase=2.65, steering_ratio=1.0):
    ws = np.array(ws, dtype=np.float64)
    lib().emu_plant_step(ws.ctypes.data_as(C.POINTER(C.c_double)), delta_cmd, thr_cmd, pitch, dtp,
                         1.0 / wheelbase, steering_ratio)
    return ws


def solve_batch(cfg, x0, p, W=None, bnd=None, traj=None, cold=True, sqp_max_iter=0, sqp_tol=1e-6, kernel="auto",
                sweeper=0):
    """cfg: any config struct with the nmpc_config field names.  sqp_max_iter = 0: one SQP-RTI
    iteration; > 0: SQP mode, at most that many iterations (KKT residuals <= sqp_tol).  kernel: "quad"
    (four instances per workgroup, nmpc_quad_solver.h), "wave" (one per workgroup, nmpc_wave_solver.h)
    or "auto" = the one-per-workgroup kernel (nmpc_solve itself picks by horizon and batch size).  Returns a dict like
    BatchSolver.get."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    B, N = x0.shape[0], cfg.N
    if W is None:
        W = np.tile(np.array(cfg.W[:]), (B, 1))
    if bnd is None:
        bnd = np.tile(np.array([cfg.lbu[0], cfg.lbu[1], cfg.lbx[0], cfg.lbx[1],
                                cfg.ubu[0], cfg.ubu[1], cfg.ubx[0], cfg.ubx[1]]), (B, 1))
    W = np.ascontiguousarray(W, dtype=np.float64)
    bnd = np.ascontiguousarray(bnd, dtype=np.float64)
    if traj is None:
        traj = np.zeros((B, N + 1, 8))
    ctrl = np.zeros((B, 6))
    st = np.zeros(B, dtype=np.int32)
    it = np.zeros(B, dtype=np.int32)
    sq = np.zeros(B, dtype=np.int32)
    cost = np.zeros(B)
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    if kernel == "auto":
        kernel = "wave"
    if kernel == "quad":
        assert sqp_max_iter == 0 and N <= 63
        rc = lib().emu_solve_batch_quad(N, cfg.dt, cfg.wheelbase, cfg.ipm_max_iter, cfg.ipm_mu0, cfg.ipm_thr0,
                                        cfg.ipm_tau, cfg.ipm_tol_mu, cfg.ipm_tol_lin, cfg.ipm_tol_comp,
                                        B, d(x0), d(p), d(W), d(bnd), d(traj), d(ctrl),
                                        i(st), i(it), i(sq), d(cost), 1 if cold else 0, sweeper,
                                        cfg.ipm_stall_iter, cfg.ipm_stall_lin)
        assert rc == 0, "emulated lanes diverged in their sync count"
        return dict(x=traj[:, :, :6].copy(), u=traj[:, :N, 6:].copy(), traj=traj, ctrl=ctrl, status=st,
                    ipm_iters=it, sqp_iters=sq, cost=cost)
    rc = lib().emu_solve_batch(N, cfg.dt, cfg.wheelbase, cfg.ipm_max_iter, cfg.ipm_mu0, cfg.ipm_thr0,
                               cfg.ipm_tau, cfg.ipm_tol_mu, cfg.ipm_tol_lin, cfg.ipm_tol_comp,
                               sqp_max_iter, sqp_tol, B, d(x0), d(p), d(W), d(bnd), d(traj), d(ctrl),
                               i(st), i(it), i(sq), d(cost), 1 if cold else 0, cfg.ipm_stall_iter, cfg.ipm_stall_lin,
                               cfg.polish, cfg.polish_mu, cfg.polish_tries, cfg.polish_tol, cfg.warm_tries, cfg.warm_gate,
                               cfg.warm_eps)
    assert rc == 0, "emulated lanes diverged in their sync count"
    return dict(x=traj[:, :, :6].copy(), u=traj[:, :N, 6:].copy(), traj=traj, ctrl=ctrl, status=st,
                ipm_iters=it, sqp_iters=sq, cost=cost)
