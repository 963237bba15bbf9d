#!/bin/bash
O=gpurun_out/ragged; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -6 $O/tests.log
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d["ms_per_step_windows"]; r=d["roofline"]; print(round(d["ms_per_step"],4), "median window", round(st.median(w),4), r["kernel"], round(r["frac"],3), r.get("avg_launch_ms"))'
echo "== config 2 (reddit shape, full cache)"; timeout 600 python bench.py --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 --skip-opt-hit --skip-cpu-baseline 2> $O/c2.err | python -c "$pick" || tail -8 $O/c2.err
echo "== config 2 graphsage"; timeout 600 python bench.py --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 --skip-opt-hit --skip-cpu-baseline --model graphsage 2> $O/c2s.err | python -c "$pick" || tail -8 $O/c2s.err
echo "== default"; timeout 600 python bench.py --skip-opt-hit --skip-cpu-baseline 2> $O/d.err | python -c "$pick" || tail -8 $O/d.err
