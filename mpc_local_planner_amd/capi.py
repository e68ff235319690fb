"""ctypes binding of the C ABI in include/nmpc.h (libnmpc_hip.so, gfx950).

There is no CPU fallback: if the HIP library is missing this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMPC_HIP_LIB selects another build of the same library (profiling variants, A/B runs)
LIB_PATH = os.environ.get("NMPC_HIP_LIB") or os.path.join(_HERE, "libnmpc_hip.so")

MODE_RTI, MODE_SQP = 0, 1
STATIC_MAX_BATCH = 16384     # NMPC_STATIC_MAX_BATCH
OK, EINVAL, EHIP, ENOMEM = 0, -1, -2, -3


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


class RunParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("loop_hz", C.c_double), ("path_tracking_vel", C.c_double),
                ("steering_ratio", C.c_double)]


EXPORTS = [
    "nmpc_default_config", "nmpc_config_from_file", "nmpc_create", "nmpc_destroy", "nmpc_set_x0", "nmpc_set_params",
    "nmpc_set_inputs_device", "nmpc_set_weights", "nmpc_set_bounds", "nmpc_set_horizons", "nmpc_reset", "nmpc_set_iterate",
    "nmpc_solve", "nmpc_solve_static",
    "nmpc_sync", "nmpc_get", "nmpc_get_view", "nmpc_get_controls", "nmpc_device_ptr", "nmpc_stream", "nmpc_set_controls_device",
    "nmpc_capacity", "nmpc_horizon", "nmpc_last_error", "nmpc_gather_controls", "nmpc_shard_range",
    "nmpc_set_track", "nmpc_frontend", "nmpc_get_frontend", "nmpc_get_world_path", "nmpc_closed_loop", "nmpc_closed_loop_measured",
]

_lib = None


class NmpcError(RuntimeError):
    pass


def lib():
    """Load libnmpc_hip.so (built in-tree by mpc_local_planner_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NmpcError(
            f"{LIB_PATH} is missing: build it with `python -m mpc_local_planner_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    l.nmpc_default_config.argtypes = [C.POINTER(Config), C.c_int, C.c_double]
    l.nmpc_config_from_file.argtypes = [C.c_char_p, C.POINTER(Config), C.POINTER(RunParams)]
    l.nmpc_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.POINTER(vp)]
    l.nmpc_destroy.argtypes = [vp]
    l.nmpc_set_x0.argtypes = [vp, dp, C.c_int]
    l.nmpc_set_params.argtypes = [vp, dp, C.c_int]
    l.nmpc_set_inputs_device.argtypes = [vp, vp, vp, C.c_int]
    l.nmpc_set_weights.argtypes = [vp, dp, C.c_int]
    l.nmpc_set_bounds.argtypes = [vp, dp, C.c_int]
    l.nmpc_set_horizons.argtypes = [vp, ip, C.c_int]
    l.nmpc_reset.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int]
    l.nmpc_set_iterate.argtypes = [vp, dp, dp, C.c_int]
    l.nmpc_solve.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double]
    l.nmpc_solve_static.argtypes = [C.POINTER(Config), C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    l.nmpc_sync.argtypes = [vp]
    l.nmpc_get.argtypes = [vp, C.c_int, dp, dp, ip, dp, ip, ip]
    l.nmpc_get_view.argtypes = [vp, C.c_int, C.POINTER(dp), C.POINTER(dp)]
    l.nmpc_get_controls.argtypes = [vp, C.c_int, dp]
    l.nmpc_device_ptr.argtypes = [vp, C.c_int]
    l.nmpc_device_ptr.restype = vp
    l.nmpc_set_controls_device.argtypes = [vp, vp]
    l.nmpc_gather_controls.argtypes = [vp, vp, C.c_int, vp]
    l.nmpc_shard_range.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    l.nmpc_stream.argtypes = [vp]
    l.nmpc_stream.restype = vp
    l.nmpc_capacity.argtypes = [vp]
    l.nmpc_horizon.argtypes = [vp]
    l.nmpc_last_error.restype = C.c_char_p
    l.nmpc_set_track.argtypes = [vp, dp, C.c_int]
    l.nmpc_frontend.argtypes = [vp, dp, dp, C.c_int]
    l.nmpc_get_frontend.argtypes = [vp, C.c_int, dp, dp, dp]
    l.nmpc_get_world_path.argtypes = [vp, C.c_int, dp]
    l.nmpc_closed_loop.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp, ip]
    l.nmpc_closed_loop_measured.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp, dp, ip]
    _lib = l
    return l


def check(rc, what):
    if rc != OK:
        raise NmpcError(f"{what} failed with code {rc}: {lib().nmpc_last_error().decode()}")


def default_config(N=40, dt=0.35, **over):
    cfg = Config()
    check(lib().nmpc_default_config(C.byref(cfg), N, dt), "nmpc_default_config")
    # NMPC_CFG="polish=0,polish_mu=1e-4": scalar overrides for A/B measurements (tools/, bench.py)
    env = {}
    for item in filter(None, os.environ.get("NMPC_CFG", "").split(",")):
        k, v = item.split("=")
        env[k.strip()] = type(getattr(cfg, k.strip()))(float(v))
    for k, v in dict(env, **over).items():
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for i, e in enumerate(v):
                cur[i] = e
        else:
            setattr(cfg, k, v)
    return cfg


def config_from_file(path):
    """(Config, RunParams) from a ROS parameter file shaped like the reference's config/mpc_params.yaml."""
    cfg, run = Config(), RunParams()
    check(lib().nmpc_config_from_file(os.fsencode(path), C.byref(cfg), C.byref(run)), "nmpc_config_from_file")
    return cfg, run


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class BatchSolver:
    """Thin object wrapper over one nmpc_handle (a batch of independent OCP instances)."""

    def __init__(self, cfg=None, capacity=1, device=0, **over):
        self.cfg = cfg if cfg is not None else default_config(**over)
        self.capacity = capacity
        self.N = self.cfg.N
        self._h = C.c_void_p()
        check(lib().nmpc_create(C.byref(self.cfg), capacity, device, C.byref(self._h)), "nmpc_create")

    def close(self):
        if self._h:
            lib().nmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_x0(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, 6)
        check(lib().nmpc_set_x0(self._h, _dp(x0), x0.shape[0]), "nmpc_set_x0")

    def set_params(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 6)
        check(lib().nmpc_set_params(self._h, _dp(p), p.shape[0]), "nmpc_set_params")

    def set_inputs_device(self, d_x0, d_p, B):
        check(lib().nmpc_set_inputs_device(self._h, d_x0, d_p, B), "nmpc_set_inputs_device")

    def set_weights(self, W=None):
        if W is None:
            check(lib().nmpc_set_weights(self._h, None, self.capacity), "nmpc_set_weights")
        else:
            W = np.ascontiguousarray(W, dtype=np.float64).reshape(-1, 7)
            check(lib().nmpc_set_weights(self._h, _dp(W), W.shape[0]), "nmpc_set_weights")

    def set_bounds(self, bnd=None):
        if bnd is None:
            check(lib().nmpc_set_bounds(self._h, None, self.capacity), "nmpc_set_bounds")
        else:
            bnd = np.ascontiguousarray(bnd, dtype=np.float64).reshape(-1, 8)
            check(lib().nmpc_set_bounds(self._h, _dp(bnd), bnd.shape[0]), "nmpc_set_bounds")

    def set_horizons(self, N=None):
        """Per-instance horizons (1..cfg.N) for the following solves, or None for cfg.N everywhere."""
        if N is None:
            check(lib().nmpc_set_horizons(self._h, None, self.capacity), "nmpc_set_horizons")
        else:
            N = np.ascontiguousarray(N, dtype=np.int32)
            check(lib().nmpc_set_horizons(self._h, _ip(N), N.shape[0]), "nmpc_set_horizons")

    def reset(self, mask=None, B=None):
        if mask is None:
            check(lib().nmpc_reset(self._h, None, B or self.capacity), "nmpc_reset")
        else:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            check(lib().nmpc_reset(self._h, mask.ctypes.data_as(C.POINTER(C.c_uint8)), mask.shape[0]),
                  "nmpc_reset")

    def set_iterate(self, x, u):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.N + 1, 6)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, self.N, 2)
        assert x.shape[0] == u.shape[0]
        check(lib().nmpc_set_iterate(self._h, _dp(x), _dp(u), x.shape[0]), "nmpc_set_iterate")

    def solve(self, B=None, mode=MODE_RTI, max_iter=1, tol=1e-6):
        check(lib().nmpc_solve(self._h, B or self.capacity, mode, max_iter, tol), "nmpc_solve")

    def sync(self):
        check(lib().nmpc_sync(self._h), "nmpc_sync")

    def get(self, B=None):
        B = B or self.capacity
        x = np.zeros((B, self.N + 1, 6))
        u = np.zeros((B, self.N, 2))
        status = np.zeros(B, dtype=np.int32)
        cost = np.zeros(B)
        it = np.zeros(B, dtype=np.int32)
        sq = np.zeros(B, dtype=np.int32)
        check(lib().nmpc_get(self._h, B, _dp(x), _dp(u), _ip(status), _dp(cost), _ip(it), _ip(sq)), "nmpc_get")
        return dict(x=x, u=u, status=status, cost=cost, ipm_iters=it, sqp_iters=sq)

    def get_view(self, B=None):
        """(x[B, N+1, 6], u[B, N, 2]) as numpy views of the handle's pinned host buffers: no copy; valid until
        the next get / get_view / close of this solver."""
        B = B or self.capacity
        px, pu = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
        check(lib().nmpc_get_view(self._h, B, C.byref(px), C.byref(pu)), "nmpc_get_view")
        x = np.ctypeslib.as_array(px, shape=(B, self.N + 1, 6))
        u = np.ctypeslib.as_array(pu, shape=(B, self.N, 2))
        return x, u

    def get_controls(self, B=None):
        B = B or self.capacity
        c = np.zeros((B, 6))
        check(lib().nmpc_get_controls(self._h, B, _dp(c)), "nmpc_get_controls")
        return c

    # ---- path front-end / closed loop
    def set_track(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        check(lib().nmpc_set_track(self._h, _dp(xy), xy.shape[0]), "nmpc_set_track")

    def frontend(self, world_state, pitch_vref):
        ws = np.ascontiguousarray(world_state, dtype=np.float64).reshape(-1, 6)
        pv = np.ascontiguousarray(pitch_vref, dtype=np.float64).reshape(-1, 2)
        assert ws.shape[0] == pv.shape[0]
        check(lib().nmpc_frontend(self._h, _dp(ws), _dp(pv), ws.shape[0]), "nmpc_frontend")

    def get_frontend(self, B=None):
        B = B or self.capacity
        x0, p, rot = np.zeros((B, 6)), np.zeros((B, 6)), np.zeros(B)
        check(lib().nmpc_get_frontend(self._h, B, _dp(x0), _dp(p), _dp(rot)), "nmpc_get_frontend")
        return x0, p, rot

    def get_world_path(self, B=None):
        B = B or self.capacity
        xy = np.zeros((B, self.N + 1, 2))
        check(lib().nmpc_get_world_path(self._h, B, _dp(xy)), "nmpc_get_world_path")
        return xy

    def closed_loop(self, world_state, pitch_vref, steps, dt_plant=1.0 / 30.0, steering_ratio=1.0,
                    trace=False, count_failures=False, steer_meas=None):
        """steer_meas[steps, B] (steering-wheel angles of the actual-steering topic, one row per tick) selects
        the use_actual_steering_topic = true branch; the return then has the published commands
        [steps, B, 2] as a fourth element."""
        ws = np.array(world_state, dtype=np.float64).reshape(-1, 6)
        pv = np.ascontiguousarray(pitch_vref, dtype=np.float64).reshape(-1, 2)
        B = ws.shape[0]
        tr = np.zeros((steps, B, 6)) if trace else None
        nf = C.c_int32(0)
        if steer_meas is not None:
            sm = np.ascontiguousarray(steer_meas, dtype=np.float64).reshape(steps, B)
            cmd = np.zeros((steps, B, 2))
            check(lib().nmpc_closed_loop_measured(self._h, B, steps, dt_plant, steering_ratio, _dp(ws), _dp(pv), _dp(sm),
                                                  _dp(cmd), _dp(tr) if trace else None,
                                                  C.byref(nf) if count_failures else None), "nmpc_closed_loop_measured")
            return ws, tr, nf.value, cmd
        check(lib().nmpc_closed_loop(self._h, B, steps, dt_plant, steering_ratio, _dp(ws), _dp(pv),
                                     _dp(tr) if trace else None, C.byref(nf) if count_failures else None),
              "nmpc_closed_loop")
        return ws, tr, nf.value

    def device_ptr(self, which):
        return lib().nmpc_device_ptr(self._h, which)

    def set_controls_device(self, ptr):
        """Device pointer (int) of a [capacity][6] float64 buffer for the control records, or None."""
        check(lib().nmpc_set_controls_device(self._h, C.c_void_p(ptr) if ptr else None), "nmpc_set_controls_device")

    def stream(self):
        return lib().nmpc_stream(self._h)
