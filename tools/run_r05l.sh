S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs"
for i in 1 2 3; do
PG_BENCH_LOSS_TRACE=1 PG_FLAT_REPLAY=1 python bench.py $S 2>&1 >/dev/null | grep -i "loss trace (epoch\|Traceback\|Error" | cut -c1-700
done
for i in 1 2; do
PG_FLAT_REPLAY=1 python bench.py $S 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('flat1', d['trained'], d['config']['epoch_ms_per_step'])"
PG_FLAT_REPLAY=0 python bench.py $S 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('flat0', d['trained'], d['config']['epoch_ms_per_step'])"
done
