# repeat one GPU test under a busy host: 14 spinning processes beside it (killed at the end)
pids=""
for i in $(seq 1 14); do python -c "
while True: pass" & pids="$pids $!"; done
fails=0
for i in $(seq 1 ${N:-10}); do
  python -m pytest tests -m gpu -x -q -k "$1" --tb=short > /tmp/flaky_$i.log 2>&1 || { fails=$((fails+1)); tail -30 /tmp/flaky_$i.log; }
done
kill $pids 2>/dev/null
echo "FAILS=$fails of ${N:-10}"
