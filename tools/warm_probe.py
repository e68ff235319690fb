import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C, torch
import bench
from mpc_local_planner_amd import capi, scenarios as sc
hip = bench.Hip(); B = 4096; N = 40
dev = torch.device("cuda", 0)
host = [sc.config2(B, start=s * B) for s in range(4)]
xb, pb = host[0]
warm = [(xb + 0.03 * s * (host[s][0] - xb), pb + 0.03 * s * (host[s][1] - pb)) for s in range(4)]
sets = [(torch.as_tensor(x).to(dev), torch.as_tensor(p).to(dev)) for x, p in warm]
torch.cuda.synchronize()
s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
for i in range(12):
    x, p = sets[i % 4]
    s.set_inputs_device(C.c_void_p(x.data_ptr()), C.c_void_p(p.data_ptr()), B)
    e0, e1 = hip.event(), hip.event()
    hip.record(e0, s.stream()); s.solve(B); hip.record(e1, s.stream())
    ms = hip.elapsed_ms(e0, e1)
    g = s.get(B)
    print(i, "ms %.3f" % ms, "ok %.4f" % (g["status"] == 0).mean(), "it mean %.2f max %d" % (g["ipm_iters"].mean(), g["ipm_iters"].max()), np.bincount(g["ipm_iters"])[6:26])
