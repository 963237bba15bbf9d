#!/bin/bash
# How many hardware queues can the process own before the side streams' kernels pay ~45 us each? (round 6)
out=${1:-gpurun_out/r06/hw_queues2.txt}; mkdir -p $(dirname $out)
base="--gpus 1 --no-configs --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --steps 1084"
: > $out
run() {
  name=$1; shift
  line=$(env "$@" python bench.py $base $FLAGS 2>/dev/null | tail -1)
  echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
q=d['ms_per_step_window_quantiles']
print('%-52s ms/step %.4f  windows p10 %.4f p50 %.4f p90 %.4f' % ('$name', d['config']['epoch_ms_per_step'], q['p10'], q['p50'], q['p90']))
" >> $out
}
FC="--cache-ratio 1.0"
FLAGS="$FC --extra-streams 1" run "full cache, one-gpu + 1 extra stream" GPU_MAX_HW_QUEUES=4
FLAGS="$FC --extra-streams 2" run "full cache, one-gpu + 2 extra streams" GPU_MAX_HW_QUEUES=4
for q in 1 2 3; do
FLAGS="$FC --extra-streams 3" run "full cache, one-gpu + 3 extra, MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q
FLAGS="$FC --dist-step" run "full cache, dist-step, MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q
FLAGS="$FC" run "full cache, one-gpu, MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q
FLAGS="" run "30% cache, one-gpu, MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q
FLAGS="--dist-step" run "30% cache, dist-step, MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q
done
FLAGS="" run "30% cache, one-gpu (runtime default: 4 queues)" GPU_MAX_HW_QUEUES=4
FLAGS="--dist-step" run "30% cache, dist-step (runtime default: 4 queues)" GPU_MAX_HW_QUEUES=4
cat $out
