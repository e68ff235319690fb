"""Measure how far a free-running SQP-RTI chain (no re-seeding) drifts from the oracle per call."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
out = {}
for N in (40, 63, 100, 127):
    B = 64
    x0, p = sc.config2(B)
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_x0(x0); s.set_params(p)
    ocfg = orc.config_like(capi.default_config(N, 0.35))
    xo = uo = None
    rows = []
    for call in range(6):
        s.solve(B); g = s.get(B)
        xprev = None if xo is None else xo.copy()
        xo, uo, so, io = orc.rti_batch(ocfg, x0, p, xo, uo, cold=(call == 0))
        ok = (so == 0) & (g["status"] == 0)
        scale = max(1.0, np.abs(xo).max(), np.abs(uo).max())
        e = np.maximum(np.abs(g["x"] - xo).reshape(B, -1).max(1), np.abs(g["u"] - uo).reshape(B, -1).max(1)) / scale
        step = np.zeros(B) if xprev is None else np.abs(xo - xprev).reshape(B, -1).max(1)
        rows.append(dict(call=call, ok=float(ok.mean()), e_max=float(e[ok].max()), e_med=float(np.median(e[ok])),
                         e_p90=float(np.quantile(e[ok], 0.9)), frac_1e6=float((e[ok] <= 1e-6).mean()),
                         frac_1e9=float((e[ok] <= 1e-9).mean()), step_med=float(np.median(step)), scale=float(scale)))
    out[N] = rows
    s.close()
print(json.dumps(out, indent=1))
