#!/bin/bash
# presample cache policy vs the number of presampled epochs; miss-path CPU gather vs threads and prefetch
set -u
OUT=${1:-gpurun_out/r04_d}
mkdir -p "$OUT"
SKIP="--skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --skip-microbench"
for ep in 1 4 16; do
  timeout 600 python bench.py --skip-cpu-baseline --skip-reference-equivalent --skip-microbench --steps 200 --no-epoch-leg --presample-epochs $ep \
      > "$OUT/bench_degree_presample_eval_ep$ep.json" 2> "$OUT/ep$ep.err"
done
timeout 600 python bench.py $SKIP --cache-policy presample --presample-epochs 16 > "$OUT/bench_presample16.json" 2>/dev/null
for th in 2 3 4 6; do
  for pf in "6,2400" "10,2400" "16,2400" "8,1024"; do
    PG_MISSQ_PREFETCH=$pf timeout 400 python bench.py $SKIP --host-threads $th --no-adapt-cpu-share > "$OUT/bench_host${th}_pf_${pf/,/_}.json" 2>/dev/null
  done
done
PG_MISSQ_PREFETCH=6,2400 timeout 400 python bench.py $SKIP --host-threads 12 --no-adapt-cpu-share > "$OUT/bench_host12_pf_6_2400.json" 2>/dev/null
PG_MISSQ_PREFETCH=6,2400 timeout 400 python bench.py $SKIP --host-threads 2 > "$OUT/bench_host2_pf_6_2400_adapt.json" 2>/dev/null
python - "$OUT" <<'PYEOF'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]; mq = d.get("miss_queue") or {}
        print(f"{os.path.basename(f):44s} ms/step {d['ms_per_step']:.4f} hit {d['cache_hit_pct_rows_fetched_by_timed_loop']:.2f}% "
              f"deg {d.get('cache_hit_degree_policy_on_trace_pct')} pre {d.get('cache_hit_presample_policy_on_trace_pct')} opt {d.get('cache_hit_oracle_upper_bound_pct')} "
              f"gather_us {mq.get('us_cpu_gather')} cpus {d['host'].get('timed_region_cgroup', {}).get('cpus_used')} share {d['config'].get('cpu_share')}")
    except Exception as e:
        print(f, "unreadable", e)
PYEOF
