"""In-tree build of the HIP library (gfx950 only).  `python -m mpc_local_planner_amd.build`."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libnmpc_hip.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -disable-machine-licm: the solve kernels sit at the 256-VGPR limit, and hoisting the materialisation
# of fp64 literals (integrator / atan polynomials) out of the SQP and slot loops only turns them into
# spills: with it the SQP kernels drop from 372 to 72 B of scratch per lane, the quad kernel from 344 to
# 132 B (+6 % throughput), the SQP-RTI wave kernels use 239 / 250 VGPRs and none (measured: DESIGN.md)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ldl", "-mllvm", "-disable-machine-licm"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in ("nmpc_hip.hip", "nmpc_wave_solver.h", "nmpc_quad_solver.h", "nmpc_quad_asm.inc", "nmpc_frontend.h",
                                             "nmpc_params_file.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "nmpc.h"))
    if force or _stale(LIB, srcs):
        cmd = [HIPCC] + FLAGS + ["-o", LIB, srcs[0]]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
