#!/bin/bash
# round 5: how the table-cached step is issued, back to back on ONE box, two alternating passes:
#   graph  = hipGraphLaunch + call-by-call prepare (rounds 2-4)      PG_FLAT_REPLAY=0 PG_NATIVE_PREPARE=0
#   tape   = the captured step as plain launches (pg_tape)           PG_NATIVE_PREPARE=0
#   both   = + prepare() as one C call (pg_batch_prepare): default
OUT=${1:-gpurun_out/replay_ab}; mkdir -p "$OUT"
S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs"
for pass in 1 2; do for v in graph tape both; do
  case $v in graph) E="PG_FLAT_REPLAY=0 PG_NATIVE_PREPARE=0";; tape) E="PG_NATIVE_PREPARE=0";; both) E="PG_X=1";; esac
  env $E python bench.py $S --cache-ratio 1.0 > "$OUT/full_cache_${v}_p$pass.json" 2>/dev/null
  env $E python bench.py $S --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 > "$OUT/config2_${v}_p$pass.json" 2>/dev/null
  env $E python bench.py $S --model graphsage --cache-ratio 1.0 > "$OUT/graphsage_full_cache_${v}_p$pass.json" 2>/dev/null
done; done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    d = json.load(open(f)); q = d["ms_per_step_window_quantiles"]
    print(f"{os.path.basename(f)[:-5]:34s} {d['config']['step_replay']:26s} ms/step {d['config']['epoch_ms_per_step']:.4f}  p50 {q['p50']:.4f} p90 {q['p90']:.4f} max {q['max']:.4f}  cpus {d['host']['timed_region_cgroup']['process_cpus_used']:.2f}  loss_last {d['trained']['loss_last']:.6f}")
PY
