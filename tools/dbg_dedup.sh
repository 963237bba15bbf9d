#!/bin/bash
O=gpurun_out/dedup; mkdir -p $O
pick='import json,sys,statistics as st; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["reference_equivalent"]; w=d["ms_per_step_windows"]; print(round(d["ms_per_step"],4), "median window", round(st.median(w),4), "min", min(w), "rows/step", round(d["miss_queue"]["timed_region"]["rows_over_pcie_per_step"]), "us gather", round(d["miss_queue"]["us_cpu_gather"]), "| ref-eq", round(r["ms_per_step"],4) if isinstance(r,dict) else r)'
B="python bench.py --steps 1084 --warmup 20 --skip-cpu-baseline --skip-microbench --skip-opt-hit"
for rep in 1 2; do for d in 1 0; do echo "== graphsage dedup=$d"; PG_DEDUP_MISSES=$d timeout 300 $B --model graphsage 2> $O/last.err | python -c "$pick" || tail -5 $O/last.err; done; done
for d in 1 0; do echo "== gcn fetch-all dedup=$d"; PG_DEDUP_MISSES=$d timeout 300 $B --fetch-all 2> $O/last.err | python -c "$pick" || tail -5 $O/last.err; done
