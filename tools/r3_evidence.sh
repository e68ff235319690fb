#!/bin/bash
# Round-3 evidence run on the GPU box: stress comparisons against the oracle (default kernel with the polish),
# the bench with the polish off on the same build, the GPU suite.  Every GPU step under its own time limit.
set -e
out=gpurun_out/$1
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/gpu_tests.log 2>&1
NMPC_CFG=polish=0 timeout -k 10 300 python bench.py > $out/bench_polish_off.json 2> $out/bench_polish_off.err
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err
STRESS_KERNELS=wave STRESS_N=40,30,63,100 timeout -k 10 900 python -u tools/stress_parity.py > $out/stress_parity.txt 2>&1
STRESS_KERNELS=wave STRESS_N=55,127 STRESS_SEEDS=30000000,40000000 timeout -k 10 900 python -u tools/stress_parity.py >> $out/stress_parity.txt 2>&1
timeout -k 10 900 python -u tools/stress_sqp.py > $out/stress_sqp.txt 2>&1
