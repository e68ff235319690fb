#!/usr/bin/env python3
"""Phase timeline of the one-instance-per-workgroup kernel under real contention: a debug build of the library
(-DNMPC_WAVE_TIMELINE) stamps the shader clock of the sweeping wave between the phases of every interior-point
iteration and of every active-set solve and sums the differences over the launch.

    python tools/wave_timeline.py build
    gpurun -- python tools/wave_timeline.py run            -> gpurun_out/wave_timeline.txt
    gpurun -- python tools/wave_timeline.py run plain      (the same launches on the shipped library: overhead of the stamps)
"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build_variants", "libnmpc_wave_timeline.so")
SRC = os.path.join(ROOT, "mpc_local_planner_amd", "csrc", "nmpc_hip.hip")
IPM = [(1, "E-A (termination tests, barrier system)"), (2, "factor sweep"), (3, "forward sweep (affine)"), (4, "E-B"),
       (5, "backward sweep"), (6, "forward sweep (corrector)"), (7, "E-C")]
POL = [(8, "pins of the working set"), (9, "factor sweep with pin elimination"), (10, "forward sweep"),
       (11, "gradient at the result"), (12, "backward sweep (multipliers)"), (13, "KKT check / correction"),
       (14, "forward sweep (refinement)"), (15, "update")]


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ldl", "-mllvm",
                           "-disable-machine-licm", "-DNMPC_WAVE_TIMELINE", "-o", LIB, SRC])
    print("built", LIB)


def run(which="timeline"):
    if which == "timeline":
        os.environ["NMPC_HIP_LIB"] = LIB
    os.environ["NMPC_KERNEL"] = "wave"
    sys.path.insert(0, ROOT)
    import numpy as np
    from mpc_local_planner_amd import capi, scenarios as sc
    lib = capi.lib()
    if which == "timeline":
        lib.nmpc_debug_wave_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    out = []
    for B, reps in ((131072, 3), (4096, 20), (256, 20)):
        x0, p = sc.config2(B)
        s = capi.BatchSolver(N=40, dt=0.35, capacity=B)
        s.set_x0(x0); s.set_params(p)
        s.solve(B); s.sync()                                  # warm-up
        if which == "timeline":
            lib.nmpc_debug_wave_timeline(None, 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            s.reset(None, B)
            s.solve(B)
        s.sync()
        wall = time.perf_counter() - t0
        g = s.get(B)
        head = (f"B = {B}, N = 40, cold RTI, {reps} launches: {1e3 * wall / reps:.3f} ms per launch "
                f"({B * reps / wall / 1e6:.2f} M solves/s), mean {g['ipm_iters'].mean():.2f} factorisations per instance")
        if which != "timeline":
            out.append("plain build, " + head)
            s.close()
            continue
        buf = (C.c_ulonglong * 24)()
        lib.nmpc_debug_wave_timeline(buf, 1)
        s.close()
        tl = np.array(buf[:], dtype=np.float64)
        n_inst, n_ipm, n_pol = tl[19], tl[17], tl[18]
        tot = tl[:17].sum()
        out.append(head + f"; per instance {n_ipm / n_inst:.2f} interior-point iterations + {n_pol / n_inst:.2f} active-set solves")
        out.append(f"  cycles per instance {tot / n_inst:10.0f}   (100 %)")
        out.append(f"  set-up: load, linearisation, tables {tl[0] / n_inst:10.0f}   ({100 * tl[0] / tot:4.1f} %)")
        out.append(f"  write-back {tl[16] / n_inst:10.0f}   ({100 * tl[16] / tot:4.1f} %)")
        ipm_tot = sum(tl[i] for i, _ in IPM)
        out.append(f"  interior-point iterations: {ipm_tot / n_ipm:8.0f} cycles each   ({100 * ipm_tot / tot:4.1f} % of the instance)")
        for i, nm in IPM:
            out.append(f"     {nm:42s} {tl[i] / n_ipm:8.0f}   ({100 * tl[i] / ipm_tot:4.1f} %)")
        pol_tot = sum(tl[i] for i, _ in POL)
        out.append(f"  active-set solves:        {pol_tot / n_pol:8.0f} cycles each   ({100 * pol_tot / tot:4.1f} % of the instance)")
        for i, nm in POL:
            out.append(f"     {nm:42s} {tl[i] / n_pol:8.0f}   ({100 * tl[i] / pol_tot:4.1f} %)")
        out.append("")
    txt = "\n".join(out)
    print(txt)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "wave_timeline.txt" if which == "timeline" else "wave_timeline_plain.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]](*sys.argv[2:])
