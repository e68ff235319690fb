#!/bin/bash
# One SQ counter pass over bench.py (LDS / VALU / wave-cycle counters of nmpc_solve_kernel).
#   gpurun -- bash tools/pmc_quick.sh <out dir under gpurun_out/>
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_quick}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/sq.log" 2>&1
echo done
