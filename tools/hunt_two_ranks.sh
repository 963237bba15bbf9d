#!/bin/bash
# the driver's N = 2 launch line with both ranks on the one GPU of the box (gloo), repeated until it hangs; keeps the hung
# run's stderr. Every run gets a session of its own and the whole process GROUP is killed at the limit (by its id).
N=${1:-10}; OUT=${2:-gpurun_out/hunt_two_ranks}; LIMIT=${3:-240}
mkdir -p "$OUT"
for i in $(seq 1 $N); do
  port=$((20000 + RANDOM % 20000))
  t0=$(date +%s)
  HSA_ENABLE_IPC_MODE_LEGACY=0 setsid python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --vertices 300000 --edges 3000000 \
      > "$OUT/run_$i.out" 2> "$OUT/run_$i.err" &
  pid=$!
  rc=hang
  for s in $(seq 1 $LIMIT); do
    if ! kill -0 $pid 2>/dev/null; then wait $pid; rc=$?; break; fi
    sleep 1
  done
  echo "run $i rc=$rc $(( $(date +%s) - t0 )) s"
  if [ "$rc" != "0" ]; then
    echo "=== stderr of run $i ([bench] lines and errors)"; grep -a "\[bench\]\|Error\|error\|Traceback\|timed out" "$OUT/run_$i.err" | tail -30 | cut -c1-300
    if [ "$rc" = "hang" ]; then
      echo "=== processes of the group"; ps -o pid,ppid,stat,etime,wchan:20,cmd -g $pid 2>/dev/null | cut -c1-200
      for p in $(pgrep -g $pid 2>/dev/null); do echo "--- $p"; cat /proc/$p/status 2>/dev/null | grep -E "State|Threads"; done
    fi
    kill -KILL -- -$pid 2>/dev/null
    break
  fi
  rm -f "$OUT/run_$i.out" "$OUT/run_$i.err"
done
