"""Config 4 timing: full SQP at N = 100 (and RTI at the same horizon for comparison)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mpc_local_planner_amd import capi, scenarios as sc
hip = bench.Hip()
N = int(os.environ.get("PROBE_N", "100")); B = 4096
x0, p = sc.config4(B)
s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
s.set_x0(x0); s.set_params(p)
for mode, kw in (("rti", {}), ("sqp20", dict(mode=capi.MODE_SQP, max_iter=20, tol=1e-6)), ("sqp1", dict(mode=capi.MODE_SQP, max_iter=1, tol=1e-6))):
    ms = []
    for _ in range(3):
        s.reset(None, B)
        e0, e1 = hip.event(), hip.event()
        hip.record(e0, s.stream()); s.solve(B, **kw); hip.record(e1, s.stream())
        ms.append(hip.elapsed_ms(e0, e1))
    g = s.get(B)
    t = float(np.median(ms))
    print(mode, dict(ms=t, qps=float(g["sqp_iters"].sum() / t * 1e3), mean_ipm=float(g["ipm_iters"].mean()), ipm_per_qp=float(g["ipm_iters"].sum() / g["sqp_iters"].sum()),
                     ok=float((g["status"] == 0).mean()), ipm_iters_per_s=float(g["ipm_iters"].sum() / t * 1e3)), flush=True)
s.close()
