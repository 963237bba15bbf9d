#!/bin/bash
# Round 4, VERDICT #1: the fused gather+aggregate kernel's own stamps against rocprofv3's dispatch times, launch by launch,
# for each way of storing the aggregated rows (PG_FWD_ROWS_STORE = plain | wt | nt). Run from the repo root on the GPU box;
# writes to gpurun_out/r04_stamps/.
set -u
OUT=${1:-gpurun_out/r04_stamps}
MODES=${MODES:-"wt plain"}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
for mode in $MODES; do
  # (a) profiled, with the one-thread marker kernel behind the fused kernel: ties the stamps' clock to the trace's
  rm -rf /tmp/prof_$mode
  ( cd /tmp && PG_FWD_ROWS_STORE=$mode PG_BENCH_STAMP_SUCCESSOR=1 PG_BENCH_DUMP_STAMPS=/tmp/stamps_$mode.npy timeout 500 \
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o b -- \
      python "$R/bench.py" $SKIP > "$R/$OUT/bench_${mode}_profiled_marker.json" 2> /tmp/prof_$mode.log )
  cp /tmp/prof_$mode/*kernel_stats.csv "$OUT/kernel_stats_${mode}_marker.csv" 2>/dev/null
  python tools/join_stamps_trace.py /tmp/stamps_$mode.npy /tmp/prof_$mode/b_kernel_trace.csv "$OUT/fused_stamps_vs_trace_${mode}_marker.csv" \
      > "$OUT/fused_stamps_vs_trace_${mode}_marker.txt" 2>&1
  python tools/trace_seq.py /tmp/prof_$mode/b_kernel_trace.csv > "$OUT/step_sequence_${mode}_marker.txt" 2>&1
  # (b) profiled without the marker: the step as shipped; the line's roofline block and the trace describe the SAME launches
  rm -rf /tmp/profn_$mode
  ( cd /tmp && PG_FWD_ROWS_STORE=$mode PG_BENCH_DUMP_STAMPS=/tmp/stampsn_$mode.npy timeout 500 \
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profn_$mode -o b -- \
      python "$R/bench.py" $SKIP > "$R/$OUT/bench_${mode}_profiled.json" 2> /tmp/profn_$mode.log )
  cp /tmp/profn_$mode/*kernel_stats.csv "$OUT/kernel_stats_${mode}.csv" 2>/dev/null
  python tools/join_stamps_trace.py /tmp/stampsn_$mode.npy /tmp/profn_$mode/b_kernel_trace.csv "$OUT/fused_stamps_vs_trace_${mode}.csv" \
      > "$OUT/fused_stamps_vs_trace_${mode}.txt" 2>&1
  python tools/trace_seq.py /tmp/profn_$mode/b_kernel_trace.csv > "$OUT/step_sequence_${mode}.txt" 2>&1
  # (c) unprofiled: 30 % cache and the table cached (compute-stream bound: where a shorter kernel shows in ms/step)
  PG_FWD_ROWS_STORE=$mode timeout 400 python bench.py $SKIP > "$OUT/bench_${mode}.json" 2>/dev/null
  PG_FWD_ROWS_STORE=$mode timeout 400 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_${mode}_full_cache.json" 2>/dev/null
done
# the kernel alone on cold rows, per store mode
for mode in plain wt nt; do
  echo "== store $mode" ; PG_FWD_ROWS_STORE=$mode timeout 300 python tools/exp_fused_rows.py 2>&1 | grep -v amdgpu.ids
done > "$OUT/fused_rows_alone_by_store.txt"
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{os.path.basename(f):40s} ms/step {d['ms_per_step']:.4f}  fused launch {r.get('avg_launch_ms', 0)*1e3:6.2f} us (body "
              f"{r.get('kernel_body_ms', 0)*1e3:6.2f}) frac {r['frac']:.3f}  loss {d['trained']['loss_first']:.3f}->{d['trained']['loss_last']:.3f}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
ls "$OUT"
