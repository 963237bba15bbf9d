#!/bin/bash
O=gpurun_out/samp; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; print(round(st.median(w),4), end=" ")'
B="python bench.py --steps 1084 --warmup 20 --skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
for v in new old; do
  if [ $v = old ]; then export PG_SAMPLER_SCAN2=1; else unset PG_SAMPLER_SCAN2; fi
  echo -n "$v gcn full: "; for i in 1 2 3; do timeout 300 $B --cache-ratio 1.0 2>/dev/null | python -c "$pick"; done; echo
  echo -n "$v config2: "; for i in 1 2; do timeout 300 python bench.py --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 --skip-opt-hit --skip-cpu-baseline 2>/dev/null | python -c "$pick"; done; echo
  echo -n "$v gcn 30%: "; for i in 1 2 3; do timeout 300 $B 2>/dev/null | python -c "$pick"; done; echo
  echo -n "$v graphsage full: "; for i in 1 2; do timeout 300 $B --model graphsage --cache-ratio 1.0 2>/dev/null | python -c "$pick"; done; echo
done
unset PG_SAMPLER_SCAN2
python tools/exp_sampler_rate.py 2>/dev/null | tail -4
