#!/bin/bash
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; print(round(st.median(w),4), end=" ")'
B="python bench.py --steps 400 --warmup 20 --skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
for ps in 3 1 6; do echo -n "gcn poll_sleeps=$ps: "; for i in 1 2 3 4; do PG_MISSQ_POLL_SLEEPS=$ps timeout 300 $B 2>/dev/null | python -c "$pick"; done; echo; done
for ps in 3 1; do echo -n "graphsage poll_sleeps=$ps: "; for i in 1 2 3; do PG_MISSQ_POLL_SLEEPS=$ps timeout 300 $B --model graphsage 2>/dev/null | python -c "$pick"; done; echo; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fetch or dedup or trainer or hardware_queue" 2>&1 | tail -1
