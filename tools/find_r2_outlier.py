"""One-off: index of the N = 127 cold instance (config 2, seed 40000000) on which the interior-point-only
kernel and oracle of round 2 ended 4.8e-4 apart (profiles/r2/stress_parity.txt:97); run with
NMPC_CFG=polish=0.  The index pins tests/test_gpu_round3.py::test_pinned_stress_outliers_are_gone."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(orc.__file__)), "libnmpc_oracle_fast.so"))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
L.orc_rti_batch_mt_h.argtypes = [C.POINTER(orc.Config), C.c_int, dp, dp, dp, dp, dp, ip, ip, C.c_int, C.c_int]
N, start, B = 127, 40_000_000, 20000
x0, p = sc.config2(B, start=start)
cfg = capi.default_config(N, 0.35)
ocfg = orc.config_like(cfg)
xo, uo = np.zeros((B, N + 1, 6)), np.zeros((B, N, 2))
so, io = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
L.orc_rti_batch_mt_h(C.byref(ocfg), B, orc._dp(x0), orc._dp(p), orc._dp(xo), orc._dp(uo), None, orc._ip(so), orc._ip(io), 1, 16)
s = capi.BatchSolver(cfg=cfg, capacity=B)
s.set_x0(x0); s.set_params(p); s.solve(B); g = s.get(B); s.close()
sc_ = np.maximum(1.0, np.maximum(np.abs(xo).reshape(B, -1).max(1), np.abs(uo).reshape(B, -1).max(1)))
e = np.maximum(np.abs(g["x"] - xo).reshape(B, -1).max(1), np.abs(g["u"] - uo).reshape(B, -1).max(1)) / sc_
print("polish", cfg.polish, "worst", e.max(), "<=1e-9", (e <= 1e-9).mean(), "<=1e-6", (e <= 1e-6).mean())
for i in np.argsort(-e)[:5]:
    print(int(i), "err %.2e" % e[i], "status", g["status"][i], so[i], "iters", g["ipm_iters"][i], io[i])
