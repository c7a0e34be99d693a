#!/bin/bash
# rocprofv3 kernel trace + instruction-mix / stall / vector-memory-path counter passes of one command, each
# pass a run of its own (counters never share a run with a trace domain, and every pass stays within the
# per-block slots: 8 SQ, 4 TCP, 2 TA).
#   tools/sq_passes.sh <out dir under gpurun_out/> <python script and its arguments ...>
set -u
out="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$out"
export TMPDIR=/tmp
CMD=("$@")
run() {   # name, rocprofv3 options...
    local name="$1"; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 "${CMD[@]}" > "$out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq_time --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA || exit 1
run pmc_sq_insts --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES || exit 1
run pmc_sq_level --pmc SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_IFETCH || exit 1
run pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum || exit 1
run pmc_lat --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum TA_TOTAL_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
python3 "$root/tools/summarize_profile.py" "$out" "$out/summary.md" > /dev/null
tail -70 "$out/summary.md"
