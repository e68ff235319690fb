#!/usr/bin/env python3
"""Phase timeline of the four-slot kernel: a debug build of the library
(-DNMPC_QUAD_TIMELINE) stamps the shader clock in every wave at each workgroup barrier and around each sweep
and sums the differences per wave index over the launch.

    python tools/quad_timeline.py build
    gpurun -- python tools/quad_timeline.py run            -> gpurun_out/quad_timeline.txt

Segments (time since the previous stamp): 0 slot work (write-back, ticket, linearisation) + E-A, 1 wait at
barrier 1, 2 factor sweep, 3 forward sweep (affine), 4 wait at barrier 2, 5 E-B, 6 wait at barrier 3,
7 backward sweep, 8 forward sweep (corrector), 9 wait at barrier 4, 10 E-C.  Wave 0 is the sweeper (and slot
0); waves 1..3 only run the element-wise phases, their segments 4 and 9 contain the sweeps of wave 0."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build_variants", "libnmpc_quad_timeline.so")
SRC = os.path.join(ROOT, "mpc_local_planner_amd", "csrc", "nmpc_hip.hip")
NAMES = ["slot work + E-A", "barrier 1 (wait)", "factor sweep", "forward sweep (affine)", "barrier 2 (wait)", "E-B",
         "barrier 3 (wait)", "backward sweep", "forward sweep (corrector)", "barrier 4 (wait)", "E-C"]


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ldl", "-mllvm",
                           "-disable-machine-licm", "-DNMPC_QUAD_TIMELINE", "-o", LIB, SRC])
    print("built", LIB)


def run(which="timeline"):
    """which = "plain": the same launches on the shipped library (no stamps), for the overhead of the stamps."""
    if which == "timeline":
        os.environ["NMPC_HIP_LIB"] = LIB
    os.environ["NMPC_KERNEL"] = "quad"
    sys.path.insert(0, ROOT)
    import numpy as np
    from mpc_local_planner_amd import capi, scenarios as sc
    lib = capi.lib()
    if which == "timeline":
        lib.nmpc_debug_quad_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    out = []
    for B, reps in ((131072, 3), (4096, 20), (256, 20)):
        x0, p = sc.config2(B)
        s = capi.BatchSolver(N=40, dt=0.35, capacity=B, polish=0)
        s.set_x0(x0); s.set_params(p)
        s.solve(B); s.sync()                                  # warm-up
        if which == "timeline":
            lib.nmpc_debug_quad_timeline(None, 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            s.reset(None, B)
            s.solve(B)
        s.sync()
        wall = time.perf_counter() - t0
        g = s.get(B)
        if which != "timeline":
            out.append(f"plain build, B = {B}: {1e3 * wall / reps:.3f} ms per launch ({B * reps / wall / 1e6:.2f} M solves/s)")
            s.close()
            continue
        buf = (C.c_ulonglong * 64)()
        lib.nmpc_debug_quad_timeline(buf, 1)
        s.close()
        tl = np.array(buf[:], dtype=np.float64).reshape(4, 16)
        iters = tl[:, 11]                                      # workgroup iterations per wave index (all equal)
        out.append(f"B = {B}, N = 40, cold RTI, {reps} launches: {1e3 * wall / reps:.3f} ms per launch "
                   f"({B * reps / wall / 1e6:.2f} M solves/s), mean {g['ipm_iters'].mean():.2f} iterations per instance, "
                   f"{iters[0] / reps:.0f} workgroup iterations per launch")
        out.append(f"{'segment':28s}" + "".join(f"{'wave ' + str(w) + (' (sweeper)' if w == 0 else ''):>22s}" for w in range(4)))
        tot = tl[:, :11].sum(1)
        for i, nm in enumerate(NAMES):
            out.append(f"{i:2d} {nm:25s}" + "".join(f"{tl[w, i] / iters[w]:13.0f} {100 * tl[w, i] / tot[w]:6.1f} %" for w in range(4)))
        out.append(f"{'   ticks per iteration':28s}" + "".join(f"{tot[w] / iters[w]:13.0f}         " for w in range(4)))
        # tick rate: the waves of a workgroup live for the whole launch when the batch is large
        out.append(f"   (sum of wave 0's ticks / workgroups resident = {tot[0] / reps:.3e} ticks per launch over all workgroups)")
        out.append("")
    txt = "\n".join(out)
    print(txt)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "quad_timeline.txt" if which == "timeline" else "quad_timeline_plain.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]](*sys.argv[2:])
