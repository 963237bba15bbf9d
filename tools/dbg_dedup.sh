#!/bin/bash
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; print(round(st.median(w),3), d["miss_queue"]["sdma_engine_mask"], end=" | ")'
B="python bench.py --gpus 1 --steps 200 --warmup 5 --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent"
for i in $(seq 1 40); do timeout 300 $B 2>/dev/null | python -c "$pick"; done; echo
echo -n "graphsage: "; for i in $(seq 1 8); do timeout 300 $B --model graphsage 2>/dev/null | python -c "$pick"; done; echo
