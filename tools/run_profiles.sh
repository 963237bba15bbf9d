#!/bin/bash
# Regenerates the files under profiles/rNN on an MI355X box (run from the repo root; writes to gpurun_out/profiles).
# Every rocprofv3 run is bounded with `timeout`; PMC passes are separate runs with --kernel-trace only.
set -u
OUT=${1:-gpurun_out/profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD

# 1. the headline line + the other bench lines
timeout 600 python bench.py > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
timeout 600 python bench.py --model graphsage --skip-opt-hit > "$OUT/bench_graphsage.json" 2>/dev/null
timeout 600 python bench.py --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 \
        --steps 260 --skip-opt-hit > "$OUT/bench_config2_reddit_shape_full_cache.json" 2>/dev/null

# 2. per-kernel time of the same command + the kernel sequence of one replayed step
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- \
      python "$R/bench.py" --skip-cpu-baseline --skip-opt-hit > /tmp/prof_stats.log 2>&1 )
cp /tmp/prof_stats/*kernel_stats.csv "$OUT/bench_kernel_stats_final.csv"
python tools/trace_seq.py /tmp/prof_stats/b_kernel_trace.csv > "$OUT/step_sequence_final.txt"

# 3. HBM bytes per kernel (FETCH_SIZE x2, WRITE_SIZE, KiB): eager loop, zero-copy misses (rocprofv3 --pmc serialises
#    all kernels; the async queue's spin-wait kernel would sit out its 3 s timeout every step)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && PG_SAMPLER_NO_GRAPH=1 timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
        python "$R/bench.py" --steps 60 --skip-cpu-baseline --skip-opt-hit --skip-microbench --no-graph \
        --miss-mode zerocopy > /tmp/pmc_$c.log 2>&1 )
done
python tools/pmc_summarize.py /tmp/pmc_FETCH_SIZE/p_counter_collection.csv /tmp/pmc_WRITE_SIZE/p_counter_collection.csv \
       "$OUT/pmc_bench_per_kernel_raw.json" > /dev/null

# 4. config 5's graph on one GPU (needs ~250 GB of host memory)
timeout 1200 python bench.py --vertices 100000000 --edges 1000000000 --steps 400 --skip-cpu-baseline --skip-opt-hit \
        > "$OUT/scale_100M_1B_single_gpu.json" 2>/dev/null
ls -la "$OUT"
