#!/bin/bash
# fused aggregate+dense (PG_FUSE_AGG_LINEAR) in the loop, full cache and 30 % cache; CU-masked side streams (PG_CU_SIDE)
set -u
OUT=${1:-gpurun_out/r04_c}
mkdir -p "$OUT"
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
for fuse in 0 1; do
  PG_FUSE_AGG_LINEAR=$fuse timeout 400 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_full_cache_fuse$fuse.json" 2>/dev/null
done
PG_FUSE_AGG_LINEAR=1 timeout 400 python bench.py $SKIP > "$OUT/bench_fuse1.json" 2>/dev/null
for side in 32 64; do
  PG_CU_SIDE=$side timeout 400 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_full_cache_side$side.json" 2> "$OUT/side$side.err"
  PG_CU_SIDE=$side timeout 400 python bench.py $SKIP > "$OUT/bench_side$side.json" 2>> "$OUT/side$side.err"
done
PG_CU_SIDE=32 PG_CU_COMPUTE_ALL=1 timeout 400 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_full_cache_side32_computeall.json" 2>/dev/null
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{os.path.basename(f):44s} ms/step {d['ms_per_step']:.4f} {r['kernel']:18s} {r.get('avg_launch_ms', 0)*1e3:6.2f} us (body {r.get('kernel_body_ms', 0)*1e3:6.2f}) frac {r['frac']:.3f} loss {d['trained']['loss_first']:.3f}->{d['trained']['loss_last']:.3f}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
tail -3 "$OUT"/side*.err
