import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
N, start, B = 127, 40_000_000, 20000
x0, p = sc.config2(B, start=start)
ocfg = orc.config_like(capi.default_config(N, 0.35))
xo, uo, so, io = orc.rti_batch(ocfg, x0, p)
s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
s.set_x0(x0); s.set_params(p); s.solve(B); g = s.get(B); s.close()
sc_ = np.maximum(1.0, np.maximum(np.abs(xo).reshape(B, -1).max(1), np.abs(uo).reshape(B, -1).max(1)))
e = np.maximum(np.abs(g["x"] - xo).reshape(B, -1).max(1), np.abs(g["u"] - uo).reshape(B, -1).max(1)) / sc_
idx = np.argsort(-e)[:5]
for i in idx:
    print(int(i), "err %.2e" % e[i], "st", g["status"][i], so[i], "it", g["ipm_iters"][i], io[i], "x0", np.array2string(x0[i], precision=4), "p", np.array2string(p[i], precision=4))
i = int(idx[0])
np.save("gpurun_out/stress_case_inputs.npy", np.concatenate([x0[i], p[i]]))
for Nn in (100, 110, 120, 121, 122, 127):
    oc = orc.config_like(capi.default_config(Nn, 0.35))
    a, b, st, it = orc.rti_batch(oc, x0[i:i+1], p[i:i+1])
    s = capi.BatchSolver(N=Nn, dt=0.35, capacity=1)
    s.set_x0(x0[i:i+1]); s.set_params(p[i:i+1]); s.solve(1); gg = s.get(1); s.close()
    print("N", Nn, "alone err %.2e" % max(np.abs(gg["x"] - a).max(), np.abs(gg["u"] - b).max()), "it", gg["ipm_iters"], it, "st", gg["status"], st)
