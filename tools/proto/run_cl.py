import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from mpc_local_planner_amd import scenarios as sc
import run_proto as rp
class Opts(C.Structure):
    _fields_ = [("method", C.c_int), ("rho", C.c_double), ("refine", C.c_int), ("pdas_max", C.c_int),
                ("tol_p", C.c_double), ("tol_d", C.c_double), ("warm_ws", C.c_int), ("mu_pol", C.c_double), ("need_stable", C.c_int),
                ("warm_eps", C.c_double), ("warm_tries", C.c_int), ("warm_gate", C.c_int)]
L = rp.L
def closed_loop(cfg, o, B=256, steps=100, seed=21):
    t = sc.test_track()
    ws, pv = sc.cars_on_track(B, t, seed=seed)
    ws = np.ascontiguousarray(ws); pv = np.ascontiguousarray(pv); t = np.ascontiguousarray(t)
    st = (rp.PStats * (B * steps))()
    L.proto_closed_loop(C.byref(cfg), C.byref(o), rp._dp(t), t.shape[0], B, rp._dp(ws), rp._dp(pv), steps, C.c_double(1/30.), st)
    S = np.array([[s.status, s.n_factor, s.n_refine, s.n_ipm, s.fallback, s.n_active] for s in st]).reshape(steps, B, 6)
    return ws, S
def rep(name, S):
    ni, nf, nr = S[..., 3], S[..., 1], S[..., 2]
    cost = ni + 0.7 * nf + 0.25 * nr
    print(f"{name}: ok {(S[...,0]==0).mean():.5f} cost/tick mean {cost.mean():.2f}; by tick [0,1,2,5,10,50,99]: "
          + " ".join(f"{cost[k].mean():.2f}/{cost[k].max():.0f}" for k in (0,1,2,5,10,50,99))
          + f" | ipm mean {ni.mean():.2f} factor {nf.mean():.2f} refine {nr.mean():.2f} warm-fail {(S[...,4]>=100).mean():.3f} nact {S[...,5].mean():.1f} max-cost {cost.max():.1f}", flush=True)
if __name__ == "__main__":
    cfg = rp.cfg_for(40)
    ws0, S0 = closed_loop(cfg, Opts(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)); rep("IPM", S0)
    for tries in (0, 1):
        for gate in (0, 1, 2, 4, 1000):
            ws2, S2 = closed_loop(cfg, Opts(2, 1e6, 2, 2, 1e-10, 1e-10, 1, 1e-3, 0, 1e-9, tries, gate)); rep(f"hybrid+warm tries {tries} gate {gate}", S2)
            d = np.abs(ws2 - ws0).max(axis=1)
            print("  end-state diff vs IPM: median", np.median(d), "99%", np.quantile(d, 0.99), "max", d.max())
