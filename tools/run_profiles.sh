#!/bin/bash
# Regenerates the files under profiles/rNN on an MI355X box (run from the repo root; writes to gpurun_out/profiles).
# Every rocprofv3 run is bounded with `timeout`; PMC passes are separate runs with --kernel-trace only.
set -u
OUT=${1:-gpurun_out/profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD

# 1. the headline line exactly as the driver runs it, the default (whole-epoch) line, and the other bench lines
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_invocation.json" 2> "$OUT/bench_driver_invocation.err"
timeout 600 python bench.py --no-configs > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
timeout 600 python bench.py --no-configs --model graphsage --skip-opt-hit > "$OUT/bench_graphsage.json" 2>/dev/null
timeout 600 python bench.py --no-configs --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 \
        --steps 260 --skip-opt-hit > "$OUT/bench_config2_reddit_shape_full_cache.json" 2>/dev/null
timeout 600 python bench.py --no-configs --no-fuse-gather --skip-opt-hit --skip-cpu-baseline > "$OUT/bench_unfused_gather.json" 2>/dev/null
timeout 600 python bench.py --no-configs --cache-ratio 1.0 --skip-opt-hit --skip-cpu-baseline > "$OUT/bench_full_cache.json" 2>/dev/null
timeout 600 python bench.py --no-configs --model graphsage --cache-ratio 1.0 --skip-opt-hit --skip-cpu-baseline > "$OUT/bench_graphsage_full_cache.json" 2>/dev/null
( timeout 300 python tools/exp_fused_rows.py; PG_FWD_ROWS_GENERIC=1 timeout 300 python tools/exp_fused_rows.py ) 2>&1 | grep -v amdgpu.ids > "$OUT/fused_rows_alone.txt"
timeout 600 python tools/exp_dup_census.py > "$OUT/dup_census.json" 2>/dev/null
timeout 300 python tools/exp_sampler_rate.py 2>&1 | grep -v amdgpu.ids > "$OUT/sampler_alone.txt"
timeout 300 python tools/exp_graph_gap.py 6 2>&1 | grep -v amdgpu.ids > "$OUT/graph_replay_gap.txt"
# the driver's N = 2 launch line, both ranks on the one GPU of the box over gloo (a path check, not a scaling number)
if [ "${SKIP_TWO_RANKS:-0}" != "1" ]; then
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --skip-cpu-baseline --skip-opt-hit \
        > "$OUT/bench_two_ranks_one_gpu_gloo.json" 2> "$OUT/bench_two_ranks_one_gpu_gloo.err"
fi

# 1b. (round 4) the table cached with block 0's aggregation inside the step (PG_EARLY_AGG=0) for the A/B with the default
PG_EARLY_AGG=0 timeout 600 python bench.py --no-configs --cache-ratio 1.0 --skip-opt-hit --skip-cpu-baseline > "$OUT/bench_full_cache_agg_in_step.json" 2>/dev/null
PG_EARLY_AGG=0 timeout 600 python bench.py --no-configs --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 \
        --steps 260 --skip-opt-hit --skip-cpu-baseline > "$OUT/bench_config2_agg_in_step.json" 2>/dev/null
timeout 600 python bench.py --no-configs --cache-policy presample --skip-cpu-baseline --skip-reference-equivalent --skip-microbench > "$OUT/bench_presample_policy.json" 2>/dev/null
timeout 600 python bench.py --no-configs --host-threads 2 --skip-opt-hit --skip-cpu-baseline --skip-reference-equivalent --skip-microbench > "$OUT/bench_host_threads_2.json" 2>/dev/null
timeout 600 python bench.py --no-configs --host-threads 4 --skip-opt-hit --skip-cpu-baseline --skip-reference-equivalent --skip-microbench > "$OUT/bench_host_threads_4.json" 2>/dev/null

# 2. per-kernel time of the same command + the kernel sequence of one replayed step; the fused kernel's own stamps of the
#    SAME launches joined with the trace (the line's roofline block must follow from the committed summary)
( cd /tmp && PG_BENCH_DUMP_STAMPS=/tmp/stamps_final.npy timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- \
      python "$R/bench.py" --no-configs --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent > "$R/$OUT/bench_profiled.json" 2> /tmp/prof_stats.log )
cp /tmp/prof_stats/*kernel_stats.csv "$OUT/bench_kernel_stats_final.csv"
python tools/trace_seq.py /tmp/prof_stats/b_kernel_trace.csv > "$OUT/step_sequence_final.txt"
python tools/join_stamps_trace.py /tmp/stamps_final.npy /tmp/prof_stats/b_kernel_trace.csv "$OUT/fused_stamps_vs_trace.csv" > "$OUT/fused_stamps_vs_trace.txt" 2>&1
python tools/trace_overlap_cond.py /tmp/prof_stats/b_kernel_trace.csv k_spmm_fwd_rows > "$OUT/fused_overlap_by_neighbour.txt" 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gs -o b -- \
      python "$R/bench.py" --no-configs --model graphsage --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench > /tmp/prof_gs.log 2>&1 )
cp /tmp/prof_gs/*kernel_stats.csv "$OUT/bench_graphsage_kernel_stats.csv"
python tools/trace_seq.py /tmp/prof_gs/b_kernel_trace.csv > "$OUT/step_sequence_graphsage.txt"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fc -o b -- \
      python "$R/bench.py" --no-configs --cache-ratio 1.0 --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench > /tmp/prof_fc.log 2>&1 )
cp /tmp/prof_fc/*kernel_stats.csv "$OUT/bench_full_cache_kernel_stats.csv"
python tools/trace_seq.py /tmp/prof_fc/b_kernel_trace.csv > "$OUT/step_sequence_full_cache.txt"

# 3. HBM bytes per kernel (FETCH_SIZE x2, WRITE_SIZE, KiB): eager loop (rocprofv3 --pmc serialises all kernels), async
#    miss queue with the consumer waiting for the worker on the host (a spin-wait kernel parked on the compute stream
#    would sit out its timeout under serialisation)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && PG_SAMPLER_NO_GRAPH=1 PG_MISSQ_HOST_WAIT=1 timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
        python "$R/bench.py" --no-configs --steps 60 --skip-cpu-baseline --skip-opt-hit --skip-microbench --skip-reference-equivalent --no-graph \
        > /tmp/pmc_$c.log 2>&1 )
done
python tools/pmc_summarize.py /tmp/pmc_FETCH_SIZE/p_counter_collection.csv /tmp/pmc_WRITE_SIZE/p_counter_collection.csv \
       "$OUT/pmc_bench_per_kernel_raw.json" > "$OUT/pmc_bench_per_kernel.txt"
python tools/pmc_aggregate.py "$OUT/pmc_bench_per_kernel_raw.json" "$OUT" >> "$OUT/pmc_bench_per_kernel.txt"   # + pmc_spmm_fwd_rows_inloop.json
# the gather micro-benchmark at 1 M rows with a calibration copy (pmc_gather_1M.md of r01)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcg_$c -o p -- \
        python "$R/tools/pmc_gather.py" > /tmp/pmcg_$c.log 2>&1 )
done
python tools/pmc_summarize.py /tmp/pmcg_FETCH_SIZE/p_counter_collection.csv /tmp/pmcg_WRITE_SIZE/p_counter_collection.csv \
       "$OUT/pmc_gather_1M_raw.json" > "$OUT/pmc_gather_1M.txt"

# 4. config 5's graph on one GPU (needs ~250 GB of host memory)
if [ "${SKIP_SCALE:-0}" != "1" ]; then
timeout 1200 python bench.py --no-configs --vertices 100000000 --edges 1000000000 --steps 400 --skip-cpu-baseline --skip-opt-hit \
        > "$OUT/scale_100M_1B_single_gpu.json" 2>/dev/null
fi
ls -la "$OUT"
