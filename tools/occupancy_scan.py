#!/usr/bin/env python3
"""Kernel time against batch size with a fixed interior-point iteration count (build_variants/
libnmpc_full10.so from tools/prof_variants.py): B = 256 is one wave per CU (pure latency),
B = 2048 fills the 8 slots per CU once.  Tells latency-bound from throughput-bound.

    python tools/prof_variants.py build && gpurun -- python tools/occupancy_scan.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "libnmpc_full10.so"
lib = os.path.join(ROOT, "build_variants", name)
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else (256, 512, 1024, 2048, 4096, 8192)
res = {}
for B in batches:
    env = dict(os.environ, NMPC_HIP_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras",
                          "--steps", "20", "--warmup", "3", "--batch", str(B)], env=env, capture_output=True,
                         text=True, timeout=200)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[B] = d["roofline"]["kernel_ms"]
    except Exception:
        res[B] = out.stderr[-300:]
    print(B, res[B], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "occupancy_scan_%s.json" % name.replace(".so", "")), "w"), indent=1)
