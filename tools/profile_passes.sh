#!/bin/bash
# rocprofv3 kernel trace + counter passes of one command, each pass a run of its own.
#   tools/profile_passes.sh <out dir under gpurun_out/> <python script and its arguments ...>
# Passes: trace (--kernel-trace --stats); pmc_fetch (FETCH_SIZE, TCC_HIT_sum); pmc_write (WRITE_SIZE,
# TCC_MISS_sum, TCC_EA0_RDREQ_DRAM_sum); pmc_req (read requests by size).  Counters never share a run
# with a trace domain.  Then tools/summarize_profile.py writes <out dir>/summary.md.
set -u
out="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$out"
export TMPDIR=/tmp
run() {   # name, rocprofv3 options...
    local name="$1"; shift
    timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 "${CMD[@]}" > "$out/$name.log" 2>&1
    echo "$name rc=$?"
}
CMD=("$@")
run trace --kernel-trace --stats || exit 1
run pmc_fetch --pmc FETCH_SIZE TCC_HIT_sum || exit 1
run pmc_write --pmc WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum || exit 1
run pmc_req --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum || exit 1
python3 "$root/tools/summarize_profile.py" "$out" "$out/summary.md" > /dev/null
tail -25 "$out/summary.md"
