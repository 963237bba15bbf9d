#!/bin/bash
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; q=d["miss_queue"]; c=d.get("miss_copy_GBps_windows") or [0]; print(round(st.median(w),3), "engine", q["sdma_engine_mask"], "copyGB/s", round(st.median(c),1), "gather", round(q["us_cpu_gather"]), "pub", round(q["us_submit_to_published"]), "done", round(q["us_submit_to_done"]), "host", round(d["host_issue_ms_per_step"],3), "cg", d["host"].get("timed_region_cgroup"))'
B="python bench.py --gpus 1 --steps 200 --warmup 5 --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent"
for i in $(seq 1 26); do PG_MISSQ_COPYLOG=1 timeout 300 $B 2>/dev/null | python -c "$pick"; done
