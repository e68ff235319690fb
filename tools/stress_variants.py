"""One-off stress of the default kernel (active-set polish on) against the oracle on problem variants the
test batches do not cover: tight bounds (many active constraints), scaled weights, short and long horizons.
Cold + one warm call, fresh instances.  Output: profiles/r3/stress_variants.txt."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc

B = int(os.environ.get("STRESS_B", "8000"))
VARIANTS = [
    ("tight bounds (u1 +-0.1, delta +-0.2)", dict(lbu=[-80, -0.1], ubu=[80, 0.1], lbx=[-0.2, -1.0], ubx=[0.2, 1.0])),
    ("very tight input rate (u0 +-2, u1 +-0.05)", dict(lbu=[-2, -0.05], ubu=[2, 0.05])),
    ("weights x10 on the path terms", dict(W=[50, 50, 10, 0.01, 0.1, 0.5, 0.1])),
    ("weights /10 on the inputs", dict(W=[5, 5, 10, 0.01, 0.1, 0.05, 0.01])),
]
for N in (40, 20, 80):
    for name, over in VARIANTS:
        x0, p = sc.config2(B, start=80_000_000 + 1000 * N)
        cfg = capi.default_config(N, 0.35, **over)
        ocfg = orc.config_like(cfg)
        hint = np.zeros(B)
        xo, uo, so, io = orc.rti_batch(ocfg, x0, p, hint=hint)
        x1 = x0 + 0.01 * (np.roll(x0, 1, axis=0) - x0)
        p1 = p + 0.01 * (np.roll(p, 1, axis=0) - p)
        xo1, uo1 = xo.copy(), uo.copy()
        xo2, uo2, so2, io2 = orc.rti_batch(ocfg, x1, p1, xo1, uo1, cold=False, hint=hint)
        failed = np.where(so != 0)[0]     # the library restarts an instance cold after a failed solve (reInit)
        if failed.size:
            xf, uf, sf, itf = orc.rti_batch(ocfg, x1[failed], p1[failed])
            xo2[failed], uo2[failed], so2[failed], io2[failed] = xf, uf, sf, itf
        s = capi.BatchSolver(cfg=cfg, capacity=B)
        s.set_x0(x0); s.set_params(p)
        s.solve(B); g = s.get(B)
        s.set_x0(x1); s.set_params(p1)
        s.solve(B); g2 = s.get(B)
        s.close()
        # (the warm call runs free on both sides: iterate and hint are each side's own)
        for call, gg, (a, b, st, it) in (("cold", g, (xo, uo, so, io)), ("warm", g2, (xo2, uo2, so2, io2))):
            ok = (st == 0) & (gg["status"] == 0)
            sc_ = np.maximum(1.0, np.maximum(np.abs(a).reshape(B, -1).max(1), np.abs(b).reshape(B, -1).max(1)))
            e = np.maximum(np.abs(gg["x"] - a).reshape(B, -1).max(1), np.abs(gg["u"] - b).reshape(B, -1).max(1)) / sc_
            print(f"N {N:3d} {name:44s} {call}: status equal {(gg['status'] == st).mean():.5f} ok {ok.mean():.5f} "
                  f"iters mean {gg['ipm_iters'].mean():.2f} (oracle {it.mean():.2f}) equal {(gg['ipm_iters'] == it).mean():.4f} "
                  f"worst {e[ok].max():.2e} <=1e-9 {(e[ok] <= 1e-9).mean():.5f} <=1e-6 {(e[ok] <= 1e-6).mean():.6f}", flush=True)
