"""One-off stress of SQP mode (all layouts) against the oracle: fresh instances, a few QPs each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
B = int(os.environ.get("STRESS_B", "2000"))
for N in (40, 55, 63, 80, 100, 127):
    for start, gen in ((60_000_000, sc.config2), (70_000_000, sc.config4)):
        x0, p = gen(B, start=start)
        s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
        s.set_x0(x0); s.set_params(p)
        s.solve(B, mode=capi.MODE_SQP, max_iter=4, tol=1e-12)
        g = s.get(B); s.close()
        ocfg = orc.config_like(capi.default_config(N, 0.35))
        def one(i):
            xo, uo = orc.init_guess(ocfg, x0[i])
            st = orc.sqp(ocfg, x0[i], p[i], xo, uo, 4, 1e-12)
            return xo, uo, st.status, st.sqp_iters
        with ThreadPoolExecutor(16) as ex:
            res = list(ex.map(one, range(B)))
        st = np.array([r[2] for r in res]); it = np.array([r[3] for r in res])
        good = (st == 0) | (st == 2)
        e = np.array([max(np.abs(g["x"][i] - res[i][0]).max(), np.abs(g["u"][i] - res[i][1]).max()) / max(1.0, np.abs(res[i][0]).max()) for i in range(B)])
        same = (g["status"] == st) & (g["sqp_iters"] == it)
        print(f"N {N:3d} {gen.__name__}: status+iters equal {same.mean():.4f} oracle ok-or-limit {good.mean():.4f} worst {e[good & same].max():.2e} "
              f"<=1e-6 {(e[good & same] <= 1e-6).mean():.5f} <=1e-9 {(e[good & same] <= 1e-9).mean():.4f}", flush=True)
