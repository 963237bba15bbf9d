mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
python tools/exp_lifetimes.py > $O/exp_lifetimes.txt 2> $O/exp_lifetimes.err; echo "lifetimes rc=$?"; cat $O/exp_lifetimes.txt
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs"
for gate in 1 0; do
  PG_PHASE_GATE=$gate python bench.py $S --cache-ratio 1.0 > $O/bench_full_cache_gate$gate.json 2> $O/bench_full_cache_gate$gate.err; echo "fullcache gate=$gate rc=$?"
  PG_PHASE_GATE=$gate python bench.py $S --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 > $O/bench_config2_gate$gate.json 2> $O/bench_config2_gate$gate.err; echo "config2 gate=$gate rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05d/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['config']['epoch_ms_per_step'],4), d['ms_per_step_window_quantiles'], round(d['host_issue_ms_per_step'],4), round(d['roofline']['frac'],3), d['roofline'].get('avg_launch_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
