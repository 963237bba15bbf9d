#!/bin/bash
# one rank's share of config 4 (dg x 4, hops 2): the bench line, then a kernel trace of the same command (dg's result is reused)
export TMPDIR=/tmp
R=$PWD
OUT=${1:-gpurun_out/r06}; mkdir -p "$OUT"
common="--gpus 1 --no-configs --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --as-rank-of 4 --dg-hops 2"
python bench.py $common > "$OUT/rank_of_4.json" 2> "$OUT/rank_of_4.err"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4 -o b -- python "$R/bench.py" $common > /tmp/prof_r4.json 2> /tmp/prof_r4.log )
cp /tmp/prof_r4/*kernel_stats.csv "$OUT/rank_of_4_kernel_stats.csv"
python tools/trace_window.py /tmp/prof_r4/b_kernel_trace.csv 100 5 > "$OUT/rank_of_4_timeline.txt"
PG_TRACE_LAUNCH=1 python bench.py $common --host-threads 4 > "$OUT/rank_of_4_threads4.json" 2>/dev/null
python bench.py $common --which-rank 3 > "$OUT/rank_of_4_rank3.json" 2>/dev/null
