#!/bin/bash
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; q=d["miss_queue"]; print(round(st.median(w),4), "gather us", round(q["us_cpu_gather"]))'
B="python bench.py --steps 600 --warmup 20 --skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
for rep in 1 2; do for cfg in "2,256" "2,2400" "3,2400" "4,2400" "1,2400" "2,1024" "4,512"; do echo -n "gcn prefetch $cfg: "; PG_MISSQ_PREFETCH=$cfg timeout 300 $B 2>/dev/null | python -c "$pick"; done; done
for cfg in "2,256" "2,2400" "4,2400"; do echo -n "graphsage prefetch $cfg: "; PG_MISSQ_PREFETCH=$cfg timeout 300 $B --model graphsage 2>/dev/null | python -c "$pick"; done
