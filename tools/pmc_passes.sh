#!/bin/bash
# PMC passes for nmpc_solve_kernel, one rocprofv3 run per pass (rocprofv3 does not split counters
# over passes; at most 8 SQ_, 4 TCC_ -- FETCH_SIZE uses 3, WRITE_SIZE 2 -- and 2 GRBM_ per run).
#   gpurun -- bash tools/pmc_passes.sh <out dir under gpurun_out/>
# then: python tools/pmc_summarize.py gpurun_out/<dir> > profiles/<round>/pmc_summary.json
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
pass() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
        python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/$name.log" 2>&1
    echo "pass $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
pass sq2 SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
