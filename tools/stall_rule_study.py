"""The stall rule (ipm_stall_iter = 20, ipm_stall_lin = 0.3: status 4 when the infeasibility scale is still above
0.3 after 20 interior-point iterations) has no counterpart in the reference's QP solver.  Oracle-only count of
QPs it turns from solved (status 0 with the rule off) into failed (status 4), over the workloads it was NOT
calibrated on: long horizons, per-instance tight bounds and swept weights, warm sequences, SQP-mode QPs.
Output: profiles/r3/stall_rule_study.txt."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_local_planner_amd import scenarios as sc
from oracle import oracle as orc

L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(orc.__file__)), "libnmpc_oracle_fast.so"))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
L.orc_rti_batch_mt_h.argtypes = [C.POINTER(orc.Config), C.c_int, dp, dp, dp, dp, dp, ip, ip, C.c_int, C.c_int]
T = int(os.environ.get("THREADS", "8"))
STALL_ITER = int(os.environ.get("STALL_ITER", "20"))     # the rule under study fires at this iteration (default: the shipped 20)


def seq(cfg, x0, p, x1, p1, calls):
    B, N = x0.shape[0], cfg.N
    xt, ut = np.zeros((B, N + 1, 6)), np.zeros((B, N, 2))
    hint = np.zeros(B)
    out = []
    for c in range(calls):
        xa, pa = x0 + 0.03 * c * (x1 - x0), p + 0.03 * c * (p1 - p)
        st, it = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        L.orc_rti_batch_mt_h(C.byref(cfg), B, orc._dp(np.ascontiguousarray(xa)), orc._dp(np.ascontiguousarray(pa)),
                             orc._dp(xt), orc._dp(ut), orc._dp(hint), orc._ip(st), orc._ip(it), 1 if c == 0 else 0, T)
        # failed instances restart cold on the next call, as the library does
        bad = st != 0
        if bad.any():
            for i in np.where(bad)[0]:
                xt[i], ut[i] = orc.init_guess(cfg, (x0 + 0.03 * (c + 1) * (x1 - x0))[i])
                hint[i] = 0
        out.append((st.copy(), it.copy()))
    return out


def study(name, N, gen, B, calls, **over):
    x0, p = gen(B)
    x1, p1 = gen(B, start=B)
    res = {}
    for stall in (STALL_ITER, 0):
        cfg = orc.default_config(N, 0.35, ipm_stall_iter=stall, **over)
        res[stall] = seq(cfg, x0, p, x1, p1, calls)
    lost = total = failed_off = 0
    for (s_on, _), (s_off, _) in zip(res[STALL_ITER], res[0]):
        lost += int(((s_on == 4) & (s_off == 0)).sum())
        failed_off += int((s_off != 0).sum())
        total += len(s_on)
    print(f"{name}: N {N} B {B} x {calls} calls: solved without the rule but failed with it: {lost} of {total} QPs; "
          f"failed without the rule: {failed_off}", flush=True)


if __name__ == "__main__":
    study("config 2 cold+warm", 40, sc.config2, 4096, 6)
    study("config 4 starts (RTI)", 100, sc.config4, 1024, 6)
    study("config 2, N = 100", 100, sc.config2, 1024, 4)
    study("config 2, N = 127", 127, sc.config2, 1024, 4)
    study("config 4, N = 127", 127, sc.config4, 512, 4)
    study("tight bounds (u1 +-0.1, delta +-0.2)", 40, sc.config2, 2048, 4, lbu=[-80, -0.1], ubu=[80, 0.1], lbx=[-0.2, -1.0], ubx=[0.2, 1.0])
    study("weights x10 on the path terms", 40, sc.config2, 2048, 4, W=[50, 50, 10, 0.01, 0.1, 0.5, 0.1])
    study("weights /10 on the inputs", 40, sc.config2, 2048, 4, W=[5, 5, 10, 0.01, 0.1, 0.05, 0.01])
    study("polish off, config 2", 40, sc.config2, 2048, 4, polish=0)
