#!/bin/bash
# end of round 4: the bench lines whose numbers the last changes touch (pg_slots_full, self-cleaning label lookup, agreed cpu-share steps)
set -u
OUT=${1:-gpurun_out/r04_final}
mkdir -p "$OUT"
S="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_invocation.json" 2> "$OUT/bench_driver_invocation.err"
timeout 300 python bench.py $S --cache-ratio 1.0 > "$OUT/bench_full_cache.json" 2>/dev/null
PG_LABELS_MEMSET=1 timeout 300 python bench.py $S --cache-ratio 1.0 > "$OUT/bench_full_cache_labels_memset.json" 2>/dev/null
timeout 300 python bench.py $S --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 > "$OUT/bench_config2_reddit_shape_full_cache.json" 2>/dev/null
timeout 300 python bench.py $S --model graphsage --cache-ratio 1.0 > "$OUT/bench_graphsage_full_cache.json" 2>/dev/null
timeout 300 python bench.py $S --model graphsage > "$OUT/bench_graphsage.json" 2>/dev/null
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f"{os.path.basename(f):48s} ms/step {d['ms_per_step']:.4f} epoch {d['config'].get('epoch_ms_per_step') or 0:.4f} value {d['value']:.4f} fused {r.get('avg_launch_ms', 0)*1e3:6.2f} us frac {r['frac']:.3f} host_issue {d.get('host_issue_ms_per_step') or 0:.4f} early {d['config'].get('early_layer0_aggregation')}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
