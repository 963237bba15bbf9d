#!/bin/bash
# Regenerates the files under profiles/r06 on an MI355X box (run from the repo root; writes to gpurun_out/profiles).
# Every rocprofv3 run is bounded with `timeout`; PMC passes are separate runs with --kernel-trace only.
#   PART=bench|trace|pmc|extra (default: all)
set -u
OUT=${1:-gpurun_out/profiles}
PART=${PART:-all}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
quick="--no-configs --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"

if [ $PART = all ] || [ $PART = bench ]; then
# 1. the headline line exactly as the driver runs it (configs 2, 3, one rank's share of 4 and 5 as child runs), and the
#    default whole-epoch line
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_invocation.json" ) 2> "$OUT/bench_driver_invocation.err"
timeout 600 python bench.py --no-configs > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
timeout 600 python bench.py $quick --cache-ratio 1.0 > "$OUT/bench_full_cache.json" 2>/dev/null
timeout 600 python bench.py $quick --model graphsage > "$OUT/bench_graphsage.json" 2>/dev/null
# the two fallback copy paths of the miss queue at the headline's size (README: the floor on the fallback)
PG_MISSQ_NO_DIRECT=1 PG_MISSQ_COPYLOG=1 timeout 600 python bench.py $quick > "$OUT/bench_missq_copy_stream.json" 2>/dev/null
PG_MISSQ_HSA_COPY=0 PG_MISSQ_COPYLOG=1 timeout 600 python bench.py $quick > "$OUT/bench_missq_hipmemcpy.json" 2>/dev/null
fi

if [ $PART = all ] || [ $PART = trace ]; then
# 2. per-kernel time of the headline command + the kernel sequence of one replayed step; the fused kernel's own stamps of
#    the SAME launches joined with the trace (the line's roofline block must follow from the committed summary)
( cd /tmp && PG_BENCH_DUMP_STAMPS=/tmp/stamps_final.npy timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- \
      python "$R/bench.py" --no-configs --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent > "$R/$OUT/bench_profiled.json" 2> /tmp/prof_stats.log )
cp /tmp/prof_stats/*kernel_stats.csv "$OUT/bench_kernel_stats_final.csv"
python tools/trace_seq.py /tmp/prof_stats/b_kernel_trace.csv > "$OUT/step_sequence_final.txt"
python tools/join_stamps_trace.py /tmp/stamps_final.npy /tmp/prof_stats/b_kernel_trace.csv "$OUT/fused_stamps_vs_trace.csv" > "$OUT/fused_stamps_vs_trace.txt" 2>&1
bash tools/prof_full_cache.sh "$OUT" full_cache > /dev/null 2>&1
fi

if [ $PART = all ] || [ $PART = pmc ]; then
# 3. HBM bytes per kernel (FETCH_SIZE x2, WRITE_SIZE, KiB): eager loop (rocprofv3 --pmc serialises all kernels), async
#    miss queue with the consumer waiting for the worker on the host (a spin-wait kernel parked on the compute stream
#    would sit out its timeout under serialisation)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && PG_MISSQ_HOST_WAIT=1 timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
        python "$R/bench.py" --no-configs --steps 60 --skip-cpu-baseline --skip-opt-hit --skip-microbench --skip-reference-equivalent --no-graph \
        > /tmp/pmc_$c.log 2>&1 )
done
python tools/pmc_summarize.py /tmp/pmc_FETCH_SIZE/p_counter_collection.csv /tmp/pmc_WRITE_SIZE/p_counter_collection.csv \
       "$OUT/pmc_bench_per_kernel_raw.json" > "$OUT/pmc_bench_per_kernel.txt"
python tools/pmc_aggregate.py "$OUT/pmc_bench_per_kernel_raw.json" "$OUT" >> "$OUT/pmc_bench_per_kernel.txt"   # + pmc_spmm_fwd_rows_inloop.json
fi

if [ $PART = all ] || [ $PART = extra ]; then
# 4. the N > 1 step shape on one GPU: A/B, the hardware-queue sweep, one rank's share of config 4 under the profiler
bash tools/ab_dist_step.sh "$OUT/ab_dist_step.txt" > /dev/null 2>&1
bash tools/exp_hw_queues2.sh "$OUT/hw_queues_sweep.txt" > /dev/null 2>&1
bash tools/prof_rank_of_4.sh "$OUT" > /dev/null 2>&1
fi
ls -la "$OUT"
