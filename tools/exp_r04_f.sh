#!/bin/bash
# early layer-0 aggregation (PG_EARLY_AGG): tests, then full cache / config 2 / 30 % cache A/B
set -u
OUT=${1:-gpurun_out/r04_f}
mkdir -p "$OUT"
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "early_layer0 or graphed_trainer or virtual_layer0 or fused_gather" 2>&1 | tail -5
for e in 0 auto; do
  for i in 1 2; do
    PG_EARLY_AGG=$e timeout 300 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_fc_early${e}_$i.json" 2>/dev/null
  done
  PG_EARLY_AGG=$e timeout 300 python bench.py $SKIP --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 > "$OUT/bench_config2_early${e}.json" 2>/dev/null
done
PG_EARLY_AGG=1 timeout 300 python bench.py $SKIP > "$OUT/bench_30pct_early1.json" 2>/dev/null
PG_EARLY_AGG=0 timeout 300 python bench.py $SKIP > "$OUT/bench_30pct_early0.json" 2>/dev/null
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f"{os.path.basename(f):36s} ms/step {d['ms_per_step']:.4f} early {d['config'].get('early_layer0_aggregation')} fused {r.get('avg_launch_ms', 0)*1e3:6.2f} us (body {r.get('kernel_body_ms', 0)*1e3:6.2f}) frac {r['frac']:.3f} loss {d['trained']['loss_first']:.3f}->{d['trained']['loss_last']:.3f} host_issue {d.get('host_issue_ms_per_step'):.4f}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
