"""One-off: closed loop of 8192 vehicles (four-slot kernel by the automatic choice) for 5 ticks against the
oracle loop of tests/test_gpu_frontend_closed_loop.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from mpc_local_planner_amd import capi
from test_frontend_emu import world_states
from test_gpu_frontend_closed_loop import _oracle_closed_loop
N, B, steps, dtp = 40, 8192, 5, 1.0 / 30.0
t, ws, pv = world_states(B, seed=77)
for kern in (None, "wave"):
    if kern: os.environ["NMPC_KERNEL"] = kern
    s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
    s.set_track(t)
    wend, trace, nfail = s.closed_loop(ws, pv, steps, dt_plant=dtp, trace=True, count_failures=True)
    s.close()
    chunks = np.array_split(np.arange(B), 16)
    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(lambda c: _oracle_closed_loop(t, ws[c], pv[c], N, steps, dtp)[1], chunks))
    to = np.concatenate(res, axis=1)
    err = np.abs(trace - to).max(axis=2)
    print(kern or "auto", "failures", nfail, "tick 1 worst %.2e" % err[0].max(), "last tick median %.2e, share <= 1e-5 %.4f, worst %.2e" % (np.median(err[-1]), (err[-1] <= 1e-5).mean(), err[-1].max()), flush=True)
