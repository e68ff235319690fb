#!/usr/bin/env python3
"""Profiling target: config 2 (B = 4096, N = 40) cold SQP-RTI steps rotating over four input sets, as
bench.py's timed loop runs them, and nothing else.  PROF_WARMUP launches first (the summarizers drop
them), then PROF_STEPS.  NMPC_KERNEL=quad profiles the four-slot kernel.  PROF_B / PROF_N override."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_local_planner_amd import capi, scenarios as sc

B = int(os.environ.get("PROF_B", "4096"))
N = int(os.environ.get("PROF_N", "40"))
W = int(os.environ.get("PROF_WARMUP", "4"))
K = int(os.environ.get("PROF_STEPS", "20"))
dev = torch.device("cuda", 0)
sets = [tuple(torch.as_tensor(a).to(dev) for a in sc.config2(B, start=s * B)) for s in range(4)]
torch.cuda.synchronize()
s = capi.BatchSolver(N=N, dt=0.35, capacity=B)
for i in range(W + K):
    x, p = sets[i % 4]
    s.set_inputs_device(C.c_void_p(x.data_ptr()), C.c_void_p(p.data_ptr()), B)
    s.reset(None, B)
    s.solve(B)
s.sync()
g = s.get(B)
print("ok", float((g["status"] == 0).mean()), float(g["ipm_iters"].mean()))
s.close()
