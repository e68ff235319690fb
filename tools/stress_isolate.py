import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
B = 20000
for N, start, bad in ((63, 20_000_000, 16768), (63, 10_000_000, 17555), (40, 20_000_000, 16768)):
    x0, p = sc.config2(B, start=start)
    ocfg = orc.config_like(capi.default_config(N, 0.35))
    xo, uo, so, io = orc.rti_batch(ocfg, x0, p)
    x1 = x0 + 0.01 * (np.roll(x0, 1, axis=0) - x0)
    p1 = p + 0.01 * (np.roll(p, 1, axis=0) - p)
    xo2, uo2, so2, io2 = orc.rti_batch(ocfg, x1, p1, xo.copy(), uo.copy(), cold=False)
    print("case", N, start, bad, "oracle warm it", io2[bad], "st", so2[bad], "neighbour", io2[bad - 1], so2[bad - 1])
    for name, sel in (("alone x1", [bad]), ("alone x4", [bad] * 4), ("with benign", [bad, 5, 6, 7]), ("with its neighbour", [bad - 1, bad, 5, 6]),
                      ("neighbour first x8", [bad - 1, 3, 4, 5, bad, 6, 7, 8])):
        sel = np.array(sel)
        for kern in ("wave", "quad"):
            os.environ["NMPC_KERNEL"] = kern
            s = capi.BatchSolver(N=N, dt=0.35, capacity=len(sel))
            s.set_iterate(xo[sel], uo[sel]); s.set_x0(x1[sel]); s.set_params(p1[sel])
            s.solve(len(sel)); g = s.get(len(sel)); s.close()
            e = np.maximum(np.abs(g["x"] - xo2[sel]).reshape(len(sel), -1).max(1), np.abs(g["u"] - uo2[sel]).reshape(len(sel), -1).max(1))
            print(f"   {name:20s} {kern}: it {g['ipm_iters'].tolist()} st {g['status'].tolist()} err {['%.1e' % v for v in e]}", flush=True)
