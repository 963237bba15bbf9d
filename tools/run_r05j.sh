mkdir -p gpurun_out/r05j; O=gpurun_out/r05j
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; echo "bench $i rc=$?"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05j/bench_driver_*.json')):
    d=json.load(open(f)); q=d['ms_per_step_window_quantiles']; print(f.split('/')[-1], 'value', round(d['value'],4), 'ms/step(20)', round(d['ms_per_step'],4), 'epoch', round(d['config']['epoch_ms_per_step'],4), q, 'frac', round(d['roofline']['frac'],3), 'cpus', d['host']['timed_region_cgroup'], {k:(round(v.get('ms_per_step',0),4)) for k,v in d['configs'].items()})
PY
grep -h "host table" $O/bench_driver_1.err
