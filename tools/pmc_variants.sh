#!/bin/bash
# One SQ counter pass per profiling variant of the library (tools/prof_variants.py build), to
# attribute LDS cycles / bank conflicts / VALU instructions to the sweeps by difference.
#   gpurun -- bash tools/pmc_variants.sh <out dir under gpurun_out/>
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_variants}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
for v in full10 nofactor nobwd nofwd nosweeps; do
    export NMPC_HIP_LIB=$ROOT/build_variants/libnmpc_$v.so
    timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU \
        --output-format csv -d "$OUT/$v" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/$v.log" 2>&1
    echo "variant $v done"
done
