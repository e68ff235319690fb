#!/usr/bin/env python3
"""Builds throw-away variants of libnmpc_hip.so with the profiling macros of
nmpc_wave_solver.h (fixed interior-point iteration count, sweeps left out) into
build_variants/, so that one gpurun call can time them side by side:

    python tools/prof_variants.py build
    gpurun -- python tools/prof_variants.py run [batch]

Results of the skip variants are numerically meaningless; only kernel time differences matter.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build_variants")
SRC = os.path.join(ROOT, "mpc_local_planner_amd", "csrc", "nmpc_hip.hip")
VARIANTS = {
    "full10": ["-DNMPC_PROF_FIXED_ITERS=10"],
    "nofactor": ["-DNMPC_PROF_FIXED_ITERS=10", "-DNMPC_PROF_SKIP=1"],
    "nobwd": ["-DNMPC_PROF_FIXED_ITERS=10", "-DNMPC_PROF_SKIP=2"],
    "nofwd": ["-DNMPC_PROF_FIXED_ITERS=10", "-DNMPC_PROF_SKIP=4"],
    "nosweeps": ["-DNMPC_PROF_FIXED_ITERS=10", "-DNMPC_PROF_SKIP=7"],
    "iters0": ["-DNMPC_PROF_FIXED_ITERS=0"],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    for name, flags in VARIANTS.items():
        lib = os.path.join(OUT, f"libnmpc_{name}.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + flags +
                              ["-o", lib, SRC])
        print("built", lib)


def run(batch="4096"):
    res = {}
    for name in VARIANTS:
        env = dict(os.environ, NMPC_HIP_LIB=os.path.join(OUT, f"libnmpc_{name}.so"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras",
                              "--steps", "20", "--warmup", "3", "--batch", batch], env=env, capture_output=True,
                             text=True, timeout=200)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[name] = d["roofline"]["kernel_ms"]
        except Exception:
            res[name] = out.stderr[-300:]
        print(name, res[name], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"prof_variants_{batch}.json"), "w"), indent=1)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]](*sys.argv[2:])
