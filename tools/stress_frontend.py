"""One-off stress of the path front-end kernel against the oracle (many vehicles around the test track)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mpc_local_planner_amd import capi
from oracle import oracle as orc
from test_frontend_emu import world_states
B = int(os.environ.get("STRESS_B", "50000"))
for seed in (101, 202):
    t, ws, pv = world_states(B, seed=seed)
    s = capi.BatchSolver(N=40, dt=0.35, capacity=B)
    s.set_track(t); s.frontend(ws, pv)
    x0s, par, rot = s.get_frontend(); s.close()
    bad_rot = bad_c = skipped = 0; worst = 0.0
    for i in range(B):
        st, en = orc.track_section(t, ws[i, 0], ws[i, 1])
        if en - st == 4:
            skipped += 1; continue
        r, c = orc.calc_coeffs(t, ws[i])
        if rot[i] != r:
            bad_rot += 1; continue
        e = np.abs(par[i, :4] - c).max() / max(1.0, np.abs(c).max())
        worst = max(worst, e); bad_c += e > 1e-9
    print(f"seed {seed}: B {B} four-point windows skipped {skipped}, rotation differs {bad_rot}, coefficients beyond 1e-9 {bad_c}, worst {worst:.2e}", flush=True)
