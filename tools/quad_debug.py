import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x0, p = sc.config2(B)
s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
s.set_x0(x0); s.set_params(p)
s.solve(B); g = s.get(B)
ocfg = orc.config_like(s.cfg)
xo, uo, so, io = orc.rti_batch(ocfg, x0, p)
print("status", g["status"], "oracle", so)
print("iters", g["ipm_iters"], "oracle", io)
e = np.maximum(np.abs(g["x"] - xo).reshape(B, -1).max(1), np.abs(g["u"] - uo).reshape(B, -1).max(1))
print("err", e)
if N <= 10 and B <= 4:
    print("x[0] gpu\n", g["x"][0]); print("x[0] orc\n", xo[0])
    print("u[0] gpu\n", g["u"][0]); print("u[0] orc\n", uo[0])
