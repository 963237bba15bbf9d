#!/bin/bash
# config 5's graph on one GPU (10^8 vertices / 10^9 edges, 240 GB host table): the bench line, and a kernel trace of the
# same run to see WHAT runs beside the fused gather+aggregate kernel there (25 us in the loop, 15.6 alone)
set -u
OUT=${1:-gpurun_out/r04_scale}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
( cd /tmp && timeout 1100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_scale -o b -- \
      python "$R/bench.py" --vertices 100000000 --edges 1000000000 --steps 400 --no-epoch-leg $SKIP > "$R/$OUT/scale_100M_1B_profiled.json" 2> /tmp/prof_scale.log )
cp /tmp/prof_scale/*kernel_stats.csv "$OUT/scale_kernel_stats.csv" 2>/dev/null
python tools/trace_seq.py /tmp/prof_scale/b_kernel_trace.csv > "$OUT/scale_step_sequence.txt" 2>&1
python tools/trace_overlap_cond.py /tmp/prof_scale/b_kernel_trace.csv k_spmm_fwd_rows > "$OUT/scale_fused_overlap.txt" 2>&1
tail -5 /tmp/prof_scale.log > "$OUT/scale_profiled.err"
cat "$OUT/scale_step_sequence.txt" "$OUT/scale_fused_overlap.txt"
head -12 "$OUT/scale_kernel_stats.csv" | cut -c1-200
