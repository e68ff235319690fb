import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from mpc_local_planner_amd import scenarios as sc
from oracle import oracle as orc

L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libproto.so"))
class Opts(C.Structure):
    _fields_ = [("method", C.c_int), ("rho", C.c_double), ("refine", C.c_int), ("pdas_max", C.c_int),
                ("tol_p", C.c_double), ("tol_d", C.c_double), ("warm_ws", C.c_int)]
class PStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("status", "n_factor", "n_refine", "n_ipm", "fallback", "n_active")]
dp = C.POINTER(C.c_double)
def _dp(a): return a.ctypes.data_as(dp)

def cfg_for(N):
    cfg = orc.Config()
    L.orc_default_config(C.byref(cfg), N, C.c_double(0.35))
    return cfg
L.orc_default_config.argtypes = [C.POINTER(orc.Config), C.c_int, C.c_double]

def run(cfg, o, x0, p, xt=None, ut=None, ws=None, cold=True):
    B = x0.shape[0]; N = cfg.N
    if xt is None:
        xt = np.zeros((B, N + 1, 6)); ut = np.zeros((B, N, 2))
    if ws is None:
        ws = np.zeros((B, N, 4), dtype=np.int32)
    st = (PStats * B)()
    L.proto_rti_batch(C.byref(cfg), C.byref(o), B, _dp(np.ascontiguousarray(x0)), _dp(np.ascontiguousarray(p)),
                      _dp(xt), _dp(ut), ws.ctypes.data_as(C.POINTER(C.c_int)), st, 1 if cold else 0)
    S = np.array([[s.status, s.n_factor, s.n_refine, s.n_ipm, s.fallback, s.n_active] for s in st])
    return xt, ut, ws, S

def summarize(name, S):
    nf, nr, ni = S[:, 1], S[:, 2], S[:, 3]
    print(f"{name}: ok {(S[:,0]==0).mean():.4f} factor mean {nf.mean():.2f} max {nf.max()} hist {np.bincount(nf)[:16]} "
          f"refine mean {nr.mean():.2f} ipm mean {ni.mean():.2f} max {ni.max()} fallback {S[:,4].mean():.4f} "
          f"active>0 {(S[:,5]>0).mean():.3f} nact mean {S[:,5].mean():.1f}", flush=True)

if __name__ == "__main__":
    B = int(os.environ.get("B", 4096)); N = int(os.environ.get("N", 40))
    gen = sc.config4 if os.environ.get("GEN") == "4" else sc.config2
    x0, p = gen(B)
    cfg = cfg_for(N)
    o_ipm = Opts(0, 0, 0, 0, 0, 0, 0)
    xi, ui, _, Si = run(cfg, o_ipm, x0, p)
    summarize("IPM cold", Si)
    for rho, refine in ((1e6, 2), (1e8, 1), (1e6, 1), (1e4, 3)):
        o = Opts(1, rho, refine, 15, 1e-10, 1e-10, 0)
        xp, up, ws, Sp = run(cfg, o, x0, p)
        summarize(f"PDAS rho {rho:g} r {refine} cold", Sp)
        ok = (Si[:, 0] == 0) & (Sp[:, 0] == 0)
        e = np.maximum(np.abs(xp - xi).reshape(B, -1).max(1), np.abs(up - ui).reshape(B, -1).max(1))
        print(f"   diff vs IPM: max {e[ok].max():.2e} 99.9% {np.quantile(e[ok], 0.999):.2e} median {np.median(e[ok]):.2e} <=1e-9 {(e[ok]<=1e-9).mean():.4f}")
    # warm sequence
    x1, p1 = gen(B, start=B)
    o = Opts(1, 1e6, 2, 15, 1e-10, 1e-10, 0)
    xp, up, ws, Sp = run(cfg, o, x0, p)
    xiw, uiw = xi.copy(), ui.copy()
    for step in range(4):
        xw, pw = x0 + 0.03 * (step + 1) * (x1 - x0), p + 0.03 * (step + 1) * (p1 - p)
        xiw, uiw, _, Siw = run(cfg, o_ipm, xw, pw, xiw, uiw, cold=False)
        summarize(f"IPM warm {step}", Siw)
        o.warm_ws = 0
        xa, ua, wsa, Sa = run(cfg, o, xw, pw, xp.copy(), up.copy(), ws.copy(), cold=False)
        summarize(f"PDAS warm {step} (ws cold)", Sa)
        o.warm_ws = 1
        xp, up, ws, Sp = run(cfg, o, xw, pw, xp, up, ws, cold=False)
        summarize(f"PDAS warm {step} (ws warm)", Sp)
        e = np.maximum(np.abs(xp - xiw).reshape(B, -1).max(1), np.abs(up - uiw).reshape(B, -1).max(1))
        print(f"   diff vs IPM chain: max {e.max():.2e} 99.9% {np.quantile(e, 0.999):.2e}")
