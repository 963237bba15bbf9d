mkdir -p gpurun_out/r03/ab
SK="--skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --host-threads 2 $SK > gpurun_out/r03/ab/copy_$i.json 2> gpurun_out/r03/ab/copy_$i.log
PG_MISSQ_TAIL_STREAM=caller python bench.py --gpus 1 --steps 20 --warmup 5 --host-threads 2 $SK > gpurun_out/r03/ab/caller_$i.json 2> gpurun_out/r03/ab/caller_$i.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/ab/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; w=d["ms_per_step_windows"]
        print(f.split("/")[-1], "epoch ms/step %.4f win %.4f share %s gather_us %.0f fused_us %.1f maxwin %.3f" % (c["epoch_ms_per_step"], d["ms_per_step"], c["cpu_share"], d["miss_queue"]["timed_region"]["us_cpu_gather"], d["roofline"]["avg_launch_ms"]*1e3, max(w)))
    except Exception as e:
        print(f, "ERR", e)
PY
