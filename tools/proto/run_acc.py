import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from mpc_local_planner_amd import scenarios as sc
import run_proto as rp
from run_cl import Opts
B, N = 2048, int(os.environ.get("N", 40))
gen = sc.config4 if os.environ.get("GEN") == "4" else sc.config2
x0, p = gen(B); x1, p1 = gen(B, start=B)
cfg = rp.cfg_for(N)
start = None
def go(o):
    xa, ua, _, Sa = rp.run(cfg, o, x0, p)
    xw, pw = x0 + 0.03 * (x1 - x0), p + 0.03 * (p1 - p)
    xs, us = (xa, ua) if start is None else start
    xb, ub, _, Sb = rp.run(cfg, o, xw, pw, xs.copy(), us.copy(), cold=False)
    return (xa, ua, Sa), (xb, ub, Sb)
gt = go(Opts(2, 1e6, 8, 2, 1e-10, 1e-10, 0, 1e-3, 0, 1e-9, 0, 0))
start = gt[0][:2]
gt = go(Opts(2, 1e6, 8, 2, 1e-10, 1e-10, 0, 1e-3, 0, 1e-9, 0, 0))
ipm = go(Opts(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
def cmp(name, r):
    out = []
    for (xa, ua, Sa), (xg, ug, Sg) in zip(r, gt):
        ok = (Sa[:, 0] == 0) & (Sg[:, 0] == 0)
        e = np.maximum(np.abs(xa - xg).reshape(B, -1).max(1), np.abs(ua - ug).reshape(B, -1).max(1))[ok]
        out.append(f"max {e.max():.1e} 99.9% {np.quantile(e, 0.999):.1e} med {np.median(e):.1e}")
    print(name, "cold:", out[0], "| warm:", out[1], flush=True)
cmp("IPM", ipm)
for rho, r in ((1e4, 6), (1e5, 2), (1e5, 3), (1e6, 1), (1e6, 2), (1e7, 1), (1e7, 2), (1e8, 1), (1e8, 2), (1e10, 0), (1e10, 1)):
    cmp(f"rho {rho:g} refine {r}", go(Opts(2, rho, r, 2, 1e-10, 1e-10, 0, 1e-3, 0, 1e-9, 0, 0)))
