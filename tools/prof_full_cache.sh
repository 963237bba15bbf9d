#!/bin/bash
# rocprofv3 kernel stats + one step's kernel sequence, table cached (the compute-stream-bound step)
export TMPDIR=/tmp
R=$PWD
OUT=${1:-gpurun_out/r06}; mkdir -p "$OUT"
TAG=${2:-full_cache}
EXTRA=${3:-}
common="--gpus 1 --no-configs --skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --cache-ratio 1.0 $EXTRA"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o b -- python "$R/bench.py" $common > "$R/$OUT/bench_${TAG}_profiled.json" 2> /tmp/prof_$TAG.log )
cp /tmp/prof_$TAG/*kernel_stats.csv "$OUT/bench_${TAG}_kernel_stats.csv"
python tools/trace_seq.py /tmp/prof_$TAG/b_kernel_trace.csv > "$OUT/step_sequence_${TAG}.txt"
python bench.py $common > "$OUT/bench_${TAG}.json" 2>/dev/null
cat "$OUT/step_sequence_${TAG}.txt"
python - <<EOF
import csv, json
rows = list(csv.DictReader(open("$OUT/bench_${TAG}_kernel_stats.csv")))
for r in rows[:16]:
    print('%-70s calls %6s avg %8.2f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
d = json.loads(open("$OUT/bench_${TAG}.json").read().strip().splitlines()[-1])
print('unprofiled ms/step', d['config']['epoch_ms_per_step'], d['ms_per_step_window_quantiles'])
EOF
