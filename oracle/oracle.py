"""ctypes binding of the CPU oracle (oracle/nmpc_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from mpc_local_planner_amd/.  Parity status:
acados parity unpinned (see nmpc_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnmpc_oracle.so")

NX, NU, NP, NY = 6, 2, 6, 7


class Config(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("dt", C.c_double), ("wheelbase", C.c_double),
        ("W", C.c_double * 7),
        ("lbu", C.c_double * 2), ("ubu", C.c_double * 2),
        ("lbx", C.c_double * 2), ("ubx", C.c_double * 2),
        ("ipm_max_iter", C.c_int), ("ipm_mu0", C.c_double), ("ipm_thr0", C.c_double),
        ("ipm_tau", C.c_double), ("ipm_tol_mu", C.c_double), ("ipm_tol_lin", C.c_double),
        ("ipm_tol_comp", C.c_double), ("ipm_stall_iter", C.c_int), ("ipm_stall_lin", C.c_double),
        ("polish", C.c_int), ("polish_mu", C.c_double), ("polish_tries", C.c_int), ("polish_tol", C.c_double),
        ("warm_tries", C.c_int), ("warm_gate", C.c_int), ("warm_eps", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("sqp_iters", C.c_int), ("ipm_iters", C.c_int),
        ("cost", C.c_double), ("res_stat", C.c_double), ("res_eq", C.c_double),
        ("res_ineq", C.c_double), ("res_comp", C.c_double), ("step_inf", C.c_double),
    ]


class QP(C.Structure):
    _fields_ = [("N", C.c_int)] + [(n, C.POINTER(C.c_double)) for n in
                                   ("A", "B", "b", "H", "g", "lu", "uu", "lx", "ux", "dx0")]


def build(force=False):
    src = os.path.join(_HERE, "nmpc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        cp = C.POINTER(Config)
        _lib.orc_default_config.argtypes = [cp, C.c_int, C.c_double]
        _lib.orc_f.argtypes = [cp, dp, dp, dp, dp]
        _lib.orc_rk4_sens.argtypes = [cp, dp, dp, dp, dp, dp, dp]
        _lib.orc_cost_yJ.argtypes = [cp, dp, dp, dp, dp, dp]
        _lib.orc_stage_gn.argtypes = [cp, dp, dp, dp, dp, dp]
        _lib.orc_qp_ipm.argtypes = [cp, C.POINTER(QP), dp, dp, dp, dp, ip, dp]
        _lib.orc_qp_ipm.restype = C.c_int
        _lib.orc_rti.argtypes = [cp, dp, dp, dp, dp, C.POINTER(Stats)]
        _lib.orc_rti.restype = C.c_int
        _lib.orc_rti_h.argtypes = [cp, dp, dp, dp, dp, dp, C.POINTER(Stats)]
        _lib.orc_rti_h.restype = C.c_int
        _lib.orc_sqp_h.argtypes = [cp, dp, dp, dp, dp, C.c_int, C.c_double, dp, C.POINTER(Stats)]
        _lib.orc_sqp_h.restype = C.c_int
        _lib.orc_qp_solve.argtypes = [cp, C.POINTER(QP), dp, dp, dp, dp, dp, ip, dp]
        _lib.orc_qp_solve.restype = C.c_int
        _lib.orc_ws_solve.argtypes = [C.POINTER(QP), dp, dp, ip, dp, dp]
        _lib.orc_ws_solve.restype = C.c_int
        _lib.orc_rti_batch_h.argtypes = [cp, C.c_int, dp, dp, dp, dp, dp, ip, ip, C.c_int]
        _lib.orc_sqp.argtypes = [cp, dp, dp, dp, dp, C.c_int, C.c_double, C.POINTER(Stats)]
        _lib.orc_sqp.restype = C.c_int
        _lib.orc_init_guess.argtypes = [cp, dp, dp, dp]
        _lib.orc_rti_batch.argtypes = [cp, C.c_int, dp, dp, dp, dp, ip, ip, C.c_int]
        _lib.orc_test_track.argtypes = [dp, C.c_int]
        _lib.orc_test_track.restype = C.c_int
        _lib.orc_track_section.argtypes = [dp, C.c_int, C.c_double, C.c_double, ip, ip]
        _lib.orc_polyfit3.argtypes = [dp, dp, C.c_int, dp]
        _lib.orc_calc_coeffs.argtypes = [dp, C.c_int, dp, dp]
        _lib.orc_calc_coeffs.restype = C.c_double
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_config(N=40, dt=0.35, **over):
    cfg = Config()
    lib().orc_default_config(C.byref(cfg), N, dt)
    for k, v in over.items():
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for i, e in enumerate(v):
                cur[i] = e
        else:
            setattr(cfg, k, v)
    return cfg


def config_like(c):
    """Oracle config with every same-named field copied from another config struct (e.g. the
    HIP library's nmpc_config), so that both sides walk the same iteration sequence."""
    cfg = Config()
    names = {n for n, _ in c._fields_}
    for n, _ in Config._fields_:
        if n not in names:
            raise KeyError(n)
        v = getattr(c, n)
        if hasattr(v, "__len__"):
            for i, e in enumerate(v):
                getattr(cfg, n)[i] = e
        else:
            setattr(cfg, n, v)
    return cfg


def f(cfg, x, u, p):
    x, u, p = _f64(x), _f64(u), _f64(p)
    out = np.zeros(NX)
    lib().orc_f(C.byref(cfg), _dp(x), _dp(u), _dp(p), _dp(out))
    return out


def rk4_sens(cfg, x, u, p):
    x, u, p = _f64(x), _f64(u), _f64(p)
    phi, A, B = np.zeros(NX), np.zeros((NX, NX)), np.zeros((NX, NU))
    lib().orc_rk4_sens(C.byref(cfg), _dp(x), _dp(u), _dp(p), _dp(phi), _dp(A), _dp(B))
    return phi, A, B


def cost_yJ(cfg, x, u, p):
    x, u, p = _f64(x), _f64(u), _f64(p)
    y, J = np.zeros(NY), np.zeros((NY, NX + NU))
    lib().orc_cost_yJ(C.byref(cfg), _dp(x), _dp(u), _dp(p), _dp(y), _dp(J))
    return y, J


def stage_gn(cfg, x, u, p):
    x, u, p = _f64(x), _f64(u), _f64(p)
    H, g = np.zeros((8, 8)), np.zeros(8)
    lib().orc_stage_gn(C.byref(cfg), _dp(x), _dp(u), _dp(p), _dp(H), _dp(g))
    return H, g


def qp_ipm(cfg, A, B, b, H, g, lu, uu, lx, ux, dx0):
    """Solve one OCP QP; arrays shaped [N,6,6],[N,6,2],[N,6],[N,8,8],[N,8],[N,2]x4,[6]."""
    N = A.shape[0]
    arrs = [_f64(a) for a in (A, B, b, H, g, lu, uu, lx, ux, dx0)]
    qp = QP(N, *[_dp(a) for a in arrs])
    dx, du = np.zeros((N + 1, NX)), np.zeros((N, NU))
    lam, pi = np.zeros((N, 8)), np.zeros((N + 1, NX))
    it = C.c_int(0)
    mu = C.c_double(0)
    st = lib().orc_qp_ipm(C.byref(cfg), C.byref(qp), _dp(dx), _dp(du), _dp(lam), _dp(pi),
                          C.byref(it), C.byref(mu))
    return dict(status=st, dx=dx, du=du, lam=lam, pi=pi, iters=it.value, mu=mu.value)


def ws_solve(A, B, b, H, g, dx0, lo, up, ws):
    """Equality-constrained QP on the working set ws[N,4] (-1 lower, +1 upper, 0 free; pairs u0,u1,x4,x5 with
    bounds lo/up[N,4]); returns (rc, z[N+1,8] per node [dx;du], sm[N,4])."""
    N = A.shape[0]
    zer = np.zeros((N, 2))
    arrs = [_f64(a) for a in (A, B, b, H, g, zer, zer, zer, zer, dx0)]
    qp = QP(N, *[_dp(a) for a in arrs])
    z = np.zeros((N + 1, 8)); z[0, :6] = dx0
    sm = np.zeros((N, 4))
    lo, up = _f64(lo), _f64(up)
    ws = np.ascontiguousarray(ws, dtype=np.int32)
    rc = lib().orc_ws_solve(C.byref(qp), _dp(lo), _dp(up), _ip(ws), _dp(z), _dp(sm))
    return rc, z, sm


def init_guess(cfg, x0):
    x0 = _f64(x0)
    xt, ut = np.zeros((cfg.N + 1, NX)), np.zeros((cfg.N, NU))
    lib().orc_init_guess(C.byref(cfg), _dp(x0), _dp(xt), _dp(ut))
    return xt, ut


def rti(cfg, x0, p, xtraj, utraj, hint=None):
    """In-place SQP-RTI step on (xtraj, utraj); returns Stats.  hint: one-element float64 array carried from
    call to call of a warm sequence (zero it on a cold start), or None."""
    x0, p = _f64(x0), _f64(p)
    assert xtraj.flags.c_contiguous and utraj.flags.c_contiguous
    st = Stats()
    lib().orc_rti_h(C.byref(cfg), _dp(x0), _dp(p), _dp(xtraj), _dp(utraj), _dp(hint) if hint is not None else None,
                    C.byref(st))
    return st


def sqp(cfg, x0, p, xtraj, utraj, max_iter=20, tol=1e-6):
    x0, p = _f64(x0), _f64(p)
    st = Stats()
    lib().orc_sqp(C.byref(cfg), _dp(x0), _dp(p), _dp(xtraj), _dp(utraj), max_iter, tol, C.byref(st))
    return st


def rti_batch(cfg, x0, p, xtraj=None, utraj=None, cold=True, hint=None):
    """hint: float64[B] warm-start hints carried from call to call (updated in place), or None."""
    x0, p = _f64(x0), _f64(p)
    Bn = x0.shape[0]
    if xtraj is None:
        xtraj = np.zeros((Bn, cfg.N + 1, NX))
        utraj = np.zeros((Bn, cfg.N, NU))
    status = np.zeros(Bn, dtype=np.int32)
    iters = np.zeros(Bn, dtype=np.int32)
    lib().orc_rti_batch_h(C.byref(cfg), Bn, _dp(x0), _dp(p), _dp(xtraj), _dp(utraj),
                          _dp(hint) if hint is not None else None, _ip(status), _ip(iters), 1 if cold else 0)
    return xtraj, utraj, status, iters


def test_track():
    buf = np.zeros((512, 2))
    n = lib().orc_test_track(_dp(buf), 512)
    return buf[:n].copy()


def track_section(track, sx, sy):
    track = _f64(track)
    s, e = C.c_int(0), C.c_int(0)
    lib().orc_track_section(_dp(track), track.shape[0], sx, sy, C.byref(s), C.byref(e))
    return s.value, e.value


def polyfit3(xs, ys):
    xs, ys = _f64(xs), _f64(ys)
    c = np.zeros(4)
    lib().orc_polyfit3(_dp(xs), _dp(ys), xs.shape[0], _dp(c))
    return c


def calc_coeffs(track, state):
    track, state = _f64(track), _f64(state)
    c = np.zeros(4)
    rot = lib().orc_calc_coeffs(_dp(track), track.shape[0], _dp(state), _dp(c))
    return rot, c
