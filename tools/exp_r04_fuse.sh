#!/bin/bash
# Round 4, VERDICT #3: layer 0's aggregation + NodeUpdate in one kernel (pg_agg_linear_fwd) vs the pair, in the loop.
set -u
OUT=${1:-gpurun_out/r04_fuse}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
for f in 1 0; do
  PG_FUSE_AGG_LINEAR=$f timeout 400 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_full_cache_fuse$f.json" 2>/dev/null
  PG_FUSE_AGG_LINEAR=$f timeout 400 python bench.py $SKIP > "$OUT/bench_fuse$f.json" 2>/dev/null
  PG_FUSE_AGG_LINEAR=$f timeout 400 python bench.py $SKIP --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 \
      --steps 260 > "$OUT/bench_config2_fuse$f.json" 2>/dev/null
  rm -rf /tmp/prof_f$f
  ( cd /tmp && PG_FUSE_AGG_LINEAR=$f PG_BENCH_DUMP_STAMPS=/tmp/stamps_f$f.npy timeout 500 rocprofv3 --kernel-trace --stats --output-format csv \
      -d /tmp/prof_f$f -o b -- python "$R/bench.py" $SKIP --cache-ratio 1.0 > "$R/$OUT/bench_full_cache_fuse${f}_profiled.json" 2> /tmp/prof_f$f.log )
  cp /tmp/prof_f$f/*kernel_stats.csv "$OUT/kernel_stats_full_cache_fuse$f.csv" 2>/dev/null
  python tools/trace_seq.py /tmp/prof_f$f/b_kernel_trace.csv > "$OUT/step_sequence_full_cache_fuse$f.txt" 2>&1
  python tools/join_stamps_trace.py /tmp/stamps_f$f.npy /tmp/prof_f$f/b_kernel_trace.csv > "$OUT/fused_stamps_vs_trace_full_cache_fuse$f.txt" 2>&1
done
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{os.path.basename(f):44s} ms/step {d['ms_per_step']:.4f} epoch {d['config']['epoch_ms_per_step']:.4f} {r['kernel']:18s} launch {r.get('avg_launch_ms', 0)*1e3:6.2f} us (body "
              f"{r.get('kernel_body_ms', 0)*1e3:6.2f}) frac {r['frac']:.3f}  loss {d['trained']['loss_first']:.3f}->{d['trained']['loss_last']:.3f}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
for f in 1 0; do echo "== fuse $f"; cat "$OUT/step_sequence_full_cache_fuse$f.txt"; done
