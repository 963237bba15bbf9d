#!/bin/bash
# full cache: what runs beside the fused kernel, and the label lookup A/B
set -u
OUT=${1:-gpurun_out/r04_e}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
for i in 1 2 3; do
  timeout 300 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_fc_block_$i.json" 2>/dev/null
  PG_LABELS_MULTIBLOCK=1 timeout 300 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_fc_multiblock_$i.json" 2>/dev/null
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fc -o b -- \
      python "$R/bench.py" --cache-ratio 1.0 $SKIP > "$R/$OUT/bench_fc_profiled.json" 2> /tmp/prof_fc.log )
cp /tmp/prof_fc/*kernel_stats.csv "$OUT/kernel_stats_full_cache.csv" 2>/dev/null
python tools/trace_seq.py /tmp/prof_fc/b_kernel_trace.csv > "$OUT/step_sequence_full_cache.txt" 2>&1
python tools/trace_overlap_cond.py /tmp/prof_fc/b_kernel_trace.csv k_spmm_fwd_rows > "$OUT/fused_overlap_full_cache.txt" 2>&1
cat "$OUT/step_sequence_full_cache.txt" "$OUT/fused_overlap_full_cache.txt"
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f"{os.path.basename(f):36s} ms/step {d['ms_per_step']:.4f} fused {r.get('avg_launch_ms', 0)*1e3:6.2f} us frac {r['frac']:.3f} {r['launch_ms_min_median_p90_max']} host_issue {d.get('host_issue_ms_per_step')}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
