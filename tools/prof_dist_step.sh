#!/bin/bash
# kernel trace of the N > 1 step shape on one GPU (table cached): per-kernel stats and the raw timeline of a few steps
export TMPDIR=/tmp
R=$PWD
OUT=${1:-gpurun_out/r06}; mkdir -p "$OUT"
common="--gpus 1 --no-configs --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --cache-ratio 1.0 --steps 600"
for leg in dist onegpu; do
  flags=""; [ $leg = dist ] && flags="--dist-step"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$leg -o b -- python "$R/bench.py" $common $flags > /tmp/prof_$leg.json 2> /tmp/prof_$leg.log )
  cp /tmp/prof_$leg/*kernel_stats.csv "$OUT/dist_step_${leg}_kernel_stats.csv"
  python tools/trace_window.py /tmp/prof_$leg/b_kernel_trace.csv 300 5 > "$OUT/dist_step_${leg}_timeline.txt"
done
