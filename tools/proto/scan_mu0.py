import sys, os, ctypes as C
import numpy as np
import run_proto as rp
from run_polish import Opts, summ, sc
B, N = 2048, 40
x0, p = sc.config2(B)
o_ipm = Opts(0, 0, 0, 0, 0, 0, 0, 0, 0)
for mu0 in (100, 10, 1, 0.1, 0.01):
    for thr0 in (0.5, 0.1, 0.02):
        cfg = rp.cfg_for(N); cfg.ipm_mu0 = mu0; cfg.ipm_thr0 = thr0
        xi, ui, _, Si = rp.run(cfg, o_ipm, x0, p)
        summ(f"mu0 {mu0} thr0 {thr0}", Si)
