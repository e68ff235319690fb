"""One-off stress: many instances (fresh seeds), both SQP-RTI kernels, cold + warm calls, against the
oracle (OpenMP build).  Prints the worst relative error and the share within 1e-9 / 1e-6."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc
import ctypes as C

lib = bench._fast_oracle()[0] if hasattr(bench, "_fast_oracle") else None
B = int(os.environ.get("STRESS_B", "20000"))
for N in [int(v) for v in os.environ.get("STRESS_N", "40,30,63,100").split(",")]:
    for start in [int(v) for v in os.environ.get("STRESS_SEEDS", "10000000,20000000").split(",")]:
        x0, p = sc.config2(B, start=start) if N != 100 else sc.config4(B // 4, start=start)
        Bn = x0.shape[0]
        x1 = x0 + 0.02 * np.roll(x0, 1, axis=0) * 0 + 0.01 * (np.roll(x0, 1, axis=0) - x0)
        p1 = p + 0.01 * (np.roll(p, 1, axis=0) - p)
        kerns = [k for k in os.environ.get("STRESS_KERNELS", "wave,quad").split(",") if k == "wave" or N <= 63]
        for kern in kerns:
            # the four-slot kernel has no active-set polish: both sides run the interior point to its own test
            over = dict(polish=0) if kern == "quad" else {}
            ocfg = orc.config_like(capi.default_config(N, 0.35, **over))
            xo, uo, so, io = orc.rti_batch(ocfg, x0, p)
            xo2, uo2, so2, io2 = orc.rti_batch(ocfg, x1, p1, xo.copy(), uo.copy(), cold=False)
            os.environ["NMPC_KERNEL"] = kern
            s = capi.BatchSolver(N=N, dt=0.35, capacity=Bn, **over)
            s.set_x0(x0); s.set_params(p)
            s.solve(Bn); g = s.get(Bn)
            s.set_iterate(xo, uo); s.set_x0(x1); s.set_params(p1)
            s.solve(Bn); g2 = s.get(Bn)
            s.close()
            for name, gg, (a, b, st) in (("cold", g, (xo, uo, so)), ("warm", g2, (xo2, uo2, so2))):
                ok = st == 0
                sc_ = np.maximum(1.0, np.maximum(np.abs(a).reshape(Bn, -1).max(1), np.abs(b).reshape(Bn, -1).max(1)))
                e = np.maximum(np.abs(gg["x"] - a).reshape(Bn, -1).max(1), np.abs(gg["u"] - b).reshape(Bn, -1).max(1)) / sc_
                same = (gg["status"] == st).mean()
                print(f"N {N:3d} seed {start} {kern} {name}: B {Bn} status equal {same:.5f} ok {ok.mean():.5f} worst {e[ok].max():.2e} "
                      f"<=1e-9 {(e[ok] <= 1e-9).mean():.5f} <=1e-6 {(e[ok] <= 1e-6).mean():.6f}", flush=True)
