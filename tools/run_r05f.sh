mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs"
for nat in 1 0; do for gate in 1 0; do
  PG_NATIVE_PREPARE=$nat PG_PHASE_GATE=$gate python bench.py $S --cache-ratio 1.0 > $O/bench_full_cache_nat${nat}_gate$gate.json 2> $O/bench_full_cache_nat${nat}_gate$gate.err; echo "fullcache nat=$nat gate=$gate rc=$?"
  PG_NATIVE_PREPARE=$nat PG_PHASE_GATE=$gate python bench.py $S --vertices 232965 --edges 57300000 --feat-size 602 --n-classes 41 --cache-ratio 1.0 --steps 260 > $O/bench_config2_nat${nat}_gate$gate.json 2> /dev/null; echo "config2 nat=$nat gate=$gate rc=$?"
done; done
PG_NATIVE_PREPARE=1 python bench.py $S --model graphsage --cache-ratio 1.0 > $O/bench_graphsage_full_cache_nat1.json 2>/dev/null
PG_NATIVE_PREPARE=0 PG_PHASE_GATE=0 python bench.py $S --model graphsage --cache-ratio 1.0 > $O/bench_graphsage_full_cache_nat0_gate0.json 2>/dev/null
timeout 900 python bench.py $S --vertices 100000000 --edges 1000000000 --steps 400 > $O/scale_100M_1B_unit_flags.json 2> $O/scale_100M_1B_unit_flags.err; echo "scale flags rc=$?"
PG_SAMPLER_NO_UNIT_FLAGS=1 timeout 900 python bench.py $S --vertices 100000000 --edges 1000000000 --steps 400 > $O/scale_100M_1B_no_unit_flags.json 2> /dev/null; echo "scale noflags rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05f/*.json')):
    try:
        d=json.load(open(f)); q=d['ms_per_step_window_quantiles']; print(f.split('/')[-1], 'ms/step', round(d['config']['epoch_ms_per_step'],4), 'p50', round(q['p50'],4), 'host', round(d['host_issue_ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'fused_us', round(1e3*(d['roofline'].get('avg_launch_ms') or 0),2))
    except Exception as e: print(f, 'ERR', e)
PY
