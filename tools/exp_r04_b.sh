#!/bin/bash
# Round 4, second batch of GPU experiments: the whole-row fused aggregate+dense kernel, the presample cache policy,
# the starved-host miss path. Run from the repo root on the GPU box; writes to gpurun_out/r04_b/.
set -u
OUT=${1:-gpurun_out/r04_b}
mkdir -p "$OUT"
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_aggregate_and_dense or presample or fused_gather_aggregate" 2>&1 | tail -5 > "$OUT/pytest_subset.txt"
cat "$OUT/pytest_subset.txt"
for fuse in 0 1; do
  PG_FUSE_AGG_LINEAR=$fuse timeout 400 python bench.py $SKIP --cache-ratio 1.0 > "$OUT/bench_full_cache_fuse$fuse.json" 2>/dev/null
  PG_FUSE_AGG_LINEAR=$fuse timeout 400 python bench.py $SKIP > "$OUT/bench_fuse$fuse.json" 2>/dev/null
done
timeout 500 python bench.py --skip-cpu-baseline --skip-reference-equivalent --skip-microbench --cache-policy presample > "$OUT/bench_presample.json" 2> "$OUT/bench_presample.err"
for pf in "2,256" "6,2400" "12,512"; do
  PG_MISSQ_PREFETCH=$pf timeout 400 python bench.py $SKIP --host-threads 2 --no-adapt-cpu-share > "$OUT/bench_host2_pf_${pf/,/_}.json" 2>/dev/null
  PG_MISSQ_PREFETCH=$pf timeout 400 python bench.py $SKIP --host-threads 4 --no-adapt-cpu-share > "$OUT/bench_host4_pf_${pf/,/_}.json" 2>/dev/null
done
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]; mq = d.get("miss_queue") or {}
        print(f"{os.path.basename(f):36s} ms/step {d['ms_per_step']:.4f} epoch {d['config']['epoch_ms_per_step']:.4f} {r['kernel']:18s} "
              f"{r.get('avg_launch_ms', 0)*1e3:6.2f} us frac {r['frac']:.3f} hit {d['cache_hit_pct_rows_fetched_by_timed_loop']:.1f}% "
              f"pre {d.get('cache_hit_presample_policy_on_trace_pct')} opt {d.get('cache_hit_oracle_upper_bound_pct')} "
              f"gather_us {mq.get('us_cpu_gather')} cpus {d['host'].get('timed_region_cgroup', {}).get('cpus_used')} loss {d['trained']['loss_first']:.3f}->{d['trained']['loss_last']:.3f}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
