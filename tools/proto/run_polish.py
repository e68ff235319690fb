import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from mpc_local_planner_amd import scenarios as sc
from oracle import oracle as orc
import run_proto as rp

class Opts(C.Structure):
    _fields_ = [("method", C.c_int), ("rho", C.c_double), ("refine", C.c_int), ("pdas_max", C.c_int),
                ("tol_p", C.c_double), ("tol_d", C.c_double), ("warm_ws", C.c_int), ("mu_pol", C.c_double), ("need_stable", C.c_int)]
rp.Opts = Opts

def summ(name, S):
    nf, nr, ni, fb = S[:, 1], S[:, 2], S[:, 3], S[:, 4]
    cost = ni + 0.7 * nf + 0.25 * nr
    print(f"{name}: ok {(S[:,0]==0).mean():.4f} ipm mean {ni.mean():.2f} max {ni.max()} hist {np.bincount(ni)[:22]} "
          f"factor {nf.mean():.2f} refine {nr.mean():.2f} failed-polish {fb.mean():.3f} max {fb.max()} cost mean {cost.mean():.2f} max {cost.max():.1f}", flush=True)

B = int(os.environ.get("B", 4096)); N = int(os.environ.get("N", 40))
gen = sc.config4 if os.environ.get("GEN") == "4" else sc.config2
x0, p = gen(B)
x1, p1 = gen(B, start=B)
cfg = rp.cfg_for(N)
o_ipm = Opts(0, 0, 0, 0, 0, 0, 0, 0, 0)
xi, ui, _, Si = rp.run(cfg, o_ipm, x0, p)
summ("IPM cold", Si)
xw, pw = x0 + 0.03 * (x1 - x0), p + 0.03 * (p1 - p)
xiw, uiw, _, Siw = rp.run(cfg, o_ipm, xw, pw, xi.copy(), ui.copy(), cold=False)
summ("IPM warm", Siw)
for mu_pol in (1.0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-6):
    for ns in (0, 1, 2):
        o = Opts(2, 1e6, 2, 0, 1e-10, 1e-10, 0, mu_pol, ns)
        xp, up, _, Sp = rp.run(cfg, o, x0, p)
        summ(f"polish mu {mu_pol:g} stable {ns} cold", Sp)
        ok = (Si[:, 0] == 0) & (Sp[:, 0] == 0)
        e = np.maximum(np.abs(xp - xi).reshape(B, -1).max(1), np.abs(up - ui).reshape(B, -1).max(1))
        print(f"   diff vs IPM: max {e[ok].max():.2e} 99.9% {np.quantile(e[ok], 0.999):.2e} <=1e-9 {(e[ok]<=1e-9).mean():.4f}")
        xp2, up2, _, Sp2 = rp.run(cfg, o, xw, pw, xp.copy(), up.copy(), cold=False)
        summ(f"polish mu {mu_pol:g} stable {ns} warm", Sp2)
