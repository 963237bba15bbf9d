mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "early_layer0 or stress or graphed_trainer" > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_subset.txt
for lag in 0 1 2; do
  PG_PHASE_LAG=$lag python bench.py $S --cache-ratio 1.0 > $O/bench_full_cache_lag$lag.json 2> /dev/null; echo "fc lag=$lag rc=$?"
  PG_PHASE_LAG=$lag python bench.py $S --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 > $O/bench_config2_lag$lag.json 2> /dev/null
  PG_PHASE_LAG=$lag python bench.py $S --model graphsage --cache-ratio 1.0 > $O/bench_graphsage_full_cache_lag$lag.json 2>/dev/null
done
PG_PHASE_GATE=0 python bench.py $S --cache-ratio 1.0 > $O/bench_full_cache_gate0.json 2> /dev/null
PG_PHASE_GATE=0 python bench.py $S --model graphsage --cache-ratio 1.0 > $O/bench_graphsage_full_cache_gate0.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05h/*.json')):
    try:
        d=json.load(open(f)); q=d['ms_per_step_window_quantiles']; print(f.split('/')[-1], 'ms/step', round(d['config']['epoch_ms_per_step'],4), 'p50', round(q['p50'],4), 'frac', round(d['roofline']['frac'],3), 'fused_us', round(1e3*(d['roofline'].get('avg_launch_ms') or 0),2), d['trained']['loss_last'])
    except Exception as e: print(f, 'ERR', e)
PY
export TMPDIR=/tmp; R=$PWD
( cd /tmp && PG_PHASE_LAG=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l0 -o b -- python $R/bench.py $S --cache-ratio 1.0 > /dev/null 2>&1 )
python tools/trace_seq.py /tmp/prof_l0/b_kernel_trace.csv
