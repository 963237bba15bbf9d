common="--gpus 1 --no-configs --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --cache-ratio 1.0 --steps 1084"
for pass in 1 2; do
for v in "default" "compute_high:PG_PRIO_COMPUTE=-1 PG_PRIO_LOAD=0 PG_PRIO_SAMPLER=0" "all_normal:PG_PRIO_COMPUTE=0 PG_PRIO_LOAD=0 PG_PRIO_SAMPLER=0" "all_high:PG_PRIO_COMPUTE=-1"; do
  name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=${v#*:}
  line=$(env $envs timeout 300 python bench.py $common 2>/dev/null | tail -1)
  echo "$line" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); q=d['ms_per_step_window_quantiles']
    print('pass $pass %-13s ms/step %.4f  p10 %.4f p50 %.4f p90 %.4f  timeouts %s' % ('$name', d['config']['epoch_ms_per_step'], q['p10'], q['p50'], q['p90'], d['config'].get('misses_timed_out')))
except Exception as e:
    print('pass $pass $name failed', e)
"
done
done
