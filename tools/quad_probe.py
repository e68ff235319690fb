"""Kernel time of a cold SQP-RTI launch over batch sizes (one process per kernel choice: NMPC_KERNEL)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mpc_local_planner_amd import capi, scenarios as sc
N = int(os.environ.get("PROBE_N", "40"))
hip = bench.Hip()
out = {}
for B in [int(b) for b in os.environ.get("PROBE_B", "4,64,256,1024,2048,4096,8192,32768,131072").split(",")]:
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    x0, p = sc.config2(B)
    s.set_x0(x0); s.set_params(p)
    s.solve(B); s.sync()
    ms = []
    for _ in range(7):
        s.reset(None, B)
        e0, e1 = hip.event(), hip.event()
        hip.record(e0, s.stream()); s.solve(B); hip.record(e1, s.stream())
        ms.append(hip.elapsed_ms(e0, e1))
    g = s.get(B)
    it = g["ipm_iters"]
    out[B] = dict(ms=float(np.median(ms)), solves_per_s=B / (np.median(ms) * 1e-3), mean_it=float(it.mean()), max_it=int(it.max()),
                  mean_max4=float(it[:B // 4 * 4].reshape(-1, 4).max(1).mean()) if B >= 4 else float(it.max()))
    s.close()
    print(B, out[B], flush=True)
