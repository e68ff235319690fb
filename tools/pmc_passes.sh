#!/bin/bash
# PMC passes and a kernel trace for the solve kernel, one rocprofv3 run per pass (rocprofv3 does not
# split counters over passes; at most 8 SQ_, 4 TCC_ -- FETCH_SIZE uses 3, WRITE_SIZE 2 -- and 2 GRBM_
# per run).  Target: tools/prof_target.py (steady-state cold steps of config 2, warm-ups first).
#   gpurun -- bash tools/pmc_passes.sh <out dir under gpurun_out/>     (NMPC_KERNEL=quad for the four-slot kernel)
# then: python tools/pmc_summarize.py gpurun_out/<dir> <kernel> 4096 <warm-up launches> > profiles/<round>/pmc_summary.json
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
pass() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- \
        python3 "$ROOT/tools/prof_target.py" > "$OUT/$name.log" 2>&1
    echo "pass $name done"
}
pass trace --kernel-trace --stats
pass fetch --pmc FETCH_SIZE
pass write --pmc WRITE_SIZE
pass sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
pass sq2 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
