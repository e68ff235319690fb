"""What a fourth resident instance per CU is worth at long horizons: N = 88 is the longest horizon
whose LDS image (54-double stages) still fits four times into a CU, N = 89 the shortest that fits
three times.  Same inputs, RTI and SQP; stage-iterations per second is the comparable figure."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mpc_local_planner_amd import capi, scenarios as sc
hip = bench.Hip()
B = int(os.environ.get("PROBE_B", "8192"))
x0, p = sc.config4(B)
for N in (60, 63, 64, 80, 88, 89, 100, 127):
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0); s.set_params(p)
    for mode, kw in (("rti", {}), ("sqp20", dict(mode=capi.MODE_SQP, max_iter=20, tol=1e-6))):
        ms = []
        for _ in range(3):
            s.reset(None, B)
            e0, e1 = hip.event(), hip.event()
            hip.record(e0, s.stream()); s.solve(B, **kw); hip.record(e1, s.stream())
            ms.append(hip.elapsed_ms(e0, e1))
        g = s.get(B)
        t = float(np.median(ms))
        it = float(g["ipm_iters"].sum())
        print(f"N {N:3d} {mode:5s} ms {t:8.3f} qp/s {g['sqp_iters'].sum() / t * 1e3:10.0f} ipm/qp {it / g['sqp_iters'].sum():5.2f} "
              f"ok {(g['status'] == 0).mean():.4f} stage-iters/s {it * N / t * 1e3:.3e}", flush=True)
    s.close()
