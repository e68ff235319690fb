"""Free-running closed loop (front-end -> SQP-RTI -> plant, 40 ticks at 30 Hz) on the GPU against the oracle loop
with its own warm-start hints: prints the world-state difference by tick and the factorisation counts of the
last tick.  gpurun -- python tools/closed_loop_parity_probe.py"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
from test_gpu_frontend_closed_loop import world_states
from test_gpu_parity import _ocfg
N, B, steps, dtp = 40, 32, 40, 1.0/30.0
t, ws, pv = world_states(B, seed=11)
s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
s.set_track(t)
wend, trace, nfail = s.closed_loop(ws, pv, steps, dt_plant=dtp, trace=True, count_failures=True)
g = s.get(B)
s.close()
pcfg = orc.default_config(N, dtp); cfg = _ocfg(N)
to = np.zeros((steps, B, 6)); its = np.zeros((steps,B),int)
for i in range(B):
    w = ws[i].copy(); xt = ut = None; hint = np.zeros(1)
    for k in range(steps):
        rot, c = orc.calc_coeffs(t, w)
        xs = np.array([0.0, 0.0, rot, w[3], w[4], w[5]]); p = np.array([c[0], c[1], c[2], c[3], pv[i, 0], pv[i, 1]])
        if xt is None: xt, ut = orc.init_guess(cfg, xs); hint[:] = 0
        st = orc.rti(cfg, xs, p, xt, ut, hint=hint); its[k,i] = st.ipm_iters
        d, th = xt[1,4], xt[1,5]
        if st.status != 0: xt = None
        x = w.copy(); x[4] = d; x[5] = th
        phi, _, _ = orc.rk4_sens(pcfg, x, np.zeros(2), np.array([0,0,0,0,pv[i,0],0.0]))
        w[:4] = phi[:4]; w[4] = d; w[5] = th
        to[k, i] = w
err = np.abs(trace - to).max(axis=2)
print("nfail", nfail, "err by tick:", [f"{err[k].max():.1e}" for k in (0,5,10,20,30,39)], "median last", np.median(err[-1]), "last-tick iters gpu", g["ipm_iters"].mean(), "oracle", its[-1].mean(), "equal", (g["ipm_iters"]==its[-1]).mean())
