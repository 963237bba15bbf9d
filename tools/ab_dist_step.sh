#!/bin/bash
# A/B of the N > 1 step shape on ONE GPU (round 6): table cached (the step is compute-stream bound there),
# two alternating passes on one box. one-gpu = the default step; dist = the N > 1 step (one-rank RCCL group) as of round 6:
# reduce-only sums into the flat buffer + in-graph all-reduce + plain Adam, ONE hipGraphLaunch (a graph that holds a collective
# is not replayed as a tape); dist-tape = the same as plain launches (tape_collectives); dist-eager = tape A, eager all-reduce,
# tape B; dist-r05 = round 5's N > 1 step (separate partial sums, AccumulateGrad adds, hipGraphLaunch).
out=${1:-gpurun_out/r06/ab_dist_step.txt}; mkdir -p $(dirname $out)
common="--gpus 1 --no-configs --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --cache-ratio 1.0 --steps 1084"
: > $out
for pass in 1 2; do
  for leg in one-gpu dist dist-tape dist-eager dist-r05; do
    case $leg in
      one-gpu) env_="" ; flags="" ;;
      dist) env_="" ; flags="--dist-step" ;;
      dist-tape) env_="" ; flags="--dist-step --tape-collectives" ;;
      dist-eager) env_="PG_GRAPH_ALLREDUCE=0" ; flags="--dist-step" ;;
      dist-r05) env_="PG_FLAT_REPLAY=0" ; flags="--dist-step --no-fuse-partials" ;;
    esac
    line=$(env $env_ python bench.py $common $flags 2>/dev/null | tail -1)
    echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
q=d['ms_per_step_window_quantiles']
print('pass $pass %-10s ms/step %.4f  windows p10 %.4f p50 %.4f p90 %.4f  replay: %s  allreduce_in_graph: %s  loss %.4f -> %.4f' % ('$leg', d['config']['epoch_ms_per_step'], q['p10'], q['p50'], q['p90'], d['config']['step_replay'], d['config']['allreduce_in_graph'], d['trained']['loss_first'], d['trained']['loss_last']))
" >> $out
  done
done
cat $out
