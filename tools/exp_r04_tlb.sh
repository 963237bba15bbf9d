#!/bin/bash
# Round 4, VERDICT #4: the fused gather+aggregate kernel over a 59.6 GB cache (config 5's shape on one GPU) vs a 6.2 GB one.
set -u
OUT=${1:-gpurun_out/r04_tlb}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translat|xnack" | head -60 > "$OUT/counters_available.txt"
timeout 900 python tools/exp_tlb.py 6.2 24 59.6 2>&1 | grep -v amdgpu.ids > "$OUT/exp_tlb.txt"
cat "$OUT/exp_tlb.txt"
