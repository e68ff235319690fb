#!/usr/bin/env python3
"""Config 5 in short form for rocprofv3 --kernel-trace --stats: 65,536 vehicles x 20 ticks.
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/closed_loop_profile.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_local_planner_amd import capi, scenarios as sc

B, ticks = 65536, int(os.environ.get("TICKS", "20"))
track = sc.test_track()
ws, pv = sc.cars_on_track(B, track)
s = capi.BatchSolver(N=40, dt=0.35, capacity=B)
s.set_track(track)
rng = np.random.default_rng(5)
s.set_weights(np.array(s.cfg.W[:])[None, :] * rng.uniform(0.5, 2.0, size=(B, 7)))
s.closed_loop(ws, pv, ticks)
s.close()
