mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "graphed_trainer or early_layer0 or stress or trainer_loop or virtual_layer0" > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_subset.txt
S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs"
for flat in 1 0; do
  PG_FLAT_REPLAY=$flat python bench.py $S --cache-ratio 1.0 > $O/bench_full_cache_flat$flat.json 2> $O/bench_full_cache_flat$flat.err; echo "fc flat=$flat rc=$?"
  PG_FLAT_REPLAY=$flat python bench.py $S --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 > $O/bench_config2_flat$flat.json 2> /dev/null
  PG_FLAT_REPLAY=$flat python bench.py $S --model graphsage --cache-ratio 1.0 > $O/bench_graphsage_full_cache_flat$flat.json 2>/dev/null
  PG_FLAT_REPLAY=$flat python bench.py $S > $O/bench_headline_flat$flat.json 2>/dev/null
  PG_FLAT_REPLAY=$flat python bench.py $S --model graphsage > $O/bench_graphsage_flat$flat.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05k/*.json')):
    try:
        d=json.load(open(f)); q=d['ms_per_step_window_quantiles']; print(f.split('/')[-1], 'ms/step', round(d['config']['epoch_ms_per_step'],4), 'p50', round(q['p50'],4), 'host', round(d['host_issue_ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'loss', d['trained']['loss_last'])
    except Exception as e: print(f, 'ERR', e)
PY
