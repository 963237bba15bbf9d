#!/bin/bash
# VERDICT r04 #1d: the GPU tier (`pytest -m gpu -x -q`, as the driver runs it) with a competing load INSIDE the box's CPU quota:
# eight busy-looping processes (8 of the 16 CPUs the cgroup may use) for the whole run.
# usage: tools/run_suite_under_load.sh [runs] [out dir] [busy processes]
N=${1:-3}; OUT=${2:-gpurun_out/under_load}; B=${3:-8}
mkdir -p "$OUT"
pids=()
for i in $(seq 1 $B); do python -c "
import time
t=time.time()
while time.time()-t < 3000: pass" & pids+=($!); done
trap 'kill "${pids[@]}" 2>/dev/null' EXIT
for i in $(seq 1 $N); do
  t0=$(date +%s)
  python -m pytest tests -m gpu -x -q > "$OUT/run_$i.txt" 2>&1; rc=$?
  echo "run $i under $B busy processes: rc=$rc $(grep -a -E '(passed|failed).* in [0-9.]+s' "$OUT/run_$i.txt" | tail -1) [$(( $(date +%s) - t0 )) s]"
  grep -a "\[bench-line\]" "$OUT/run_$i.txt" | cut -c1-330
  [ $rc -ne 0 ] && grep -a -n "FAILED\|\[conftest\]\|Error" "$OUT/run_$i.txt" | head -8 | cut -c1-300
done
kill "${pids[@]}" 2>/dev/null
