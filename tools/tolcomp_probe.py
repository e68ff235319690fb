"""Sensitivity of the launch time to the pairwise-complementarity tolerance (nmpc_config.ipm_tol_comp):
cold and warm SQP-RTI at config 2; error is measured against the default (1e-11) result."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mpc_local_planner_amd import capi, scenarios as sc
hip = bench.Hip()
B, N = 4096, 40
x0, p = sc.config2(B)
x1, p1 = sc.config2(B, start=B)
xw, pw = x0 + 0.03 * (x1 - x0), p + 0.03 * (p1 - p)
ref = {}
for tol in (1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-4, 1.0):
    cfg = capi.default_config(N, 0.35)
    cfg.ipm_tol_comp = tol
    s = capi.BatchSolver(cfg=cfg, capacity=B) if "cfg" in capi.BatchSolver.__init__.__code__.co_varnames else None
    if s is None:
        raise SystemExit("BatchSolver has no cfg argument")
    out = {}
    for name, (a, b), cold in (("cold", (x0, p), True), ("warm", (xw, pw), False)):
        ms = []
        for rep in range(5):
            s.set_x0(x0); s.set_params(p); s.reset(None, B); s.solve(B); s.sync()      # cold start point
            if not cold:
                s.set_x0(a); s.set_params(b)
            else:
                s.reset(None, B)
            e0, e1 = hip.event(), hip.event()
            hip.record(e0, s.stream()); s.solve(B); hip.record(e1, s.stream())
            ms.append(hip.elapsed_ms(e0, e1))
        g = s.get(B)
        if tol == 1e-11:
            ref[name] = g
        err = max(np.abs(g["x"] - ref[name]["x"]).max(), np.abs(g["u"] - ref[name]["u"]).max())
        it = g["ipm_iters"]
        out[name] = f"{name} ms {np.median(ms):.3f} it mean {it.mean():.2f} max {it.max()} ok {(g['status'] == 0).mean():.4f} err vs 1e-11 {err:.2e}"
    print(f"tol_comp {tol:g}: " + " | ".join(out.values()), flush=True)
    s.close()
