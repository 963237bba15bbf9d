mkdir -p gpurun_out/r03/tail2
SK="--skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
for nb in 4 8 12 16 24 48; do
  PG_SCATTER_HOST_BLOCKS=$nb python bench.py --gpus 1 --steps 20 --warmup 5 --host-threads 2 --cpu-share 0.0 $SK > gpurun_out/r03/tail2/wide_s0_b$nb.json 2> gpurun_out/r03/tail2/wide_s0_b$nb.log
done
for nb in 8 16 24; do
  PG_SCATTER_HOST_BLOCKS=$nb python bench.py --gpus 1 --steps 20 --warmup 5 --host-threads 2 --cpu-share 0.3125 $SK > gpurun_out/r03/tail2/wide_s31_b$nb.json 2> gpurun_out/r03/tail2/wide_s31_b$nb.log
  PG_SCATTER_HOST_NARROW=1 PG_SCATTER_HOST_BLOCKS=$nb python bench.py --gpus 1 --steps 20 --warmup 5 --host-threads 2 --cpu-share 0.0 $SK > gpurun_out/r03/tail2/narrow_s0_b$nb.json 2> gpurun_out/r03/tail2/narrow_s0_b$nb.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/tail2/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; w=d["ms_per_step_windows"]
        print(f.split("/")[-1], "epoch ms/step %.4f win %.4f share %s gather_us %.0f fused_us %.1f maxwin %.3f" % (c["epoch_ms_per_step"], d["ms_per_step"], c["cpu_share"], d["miss_queue"]["timed_region"]["us_cpu_gather"], d["roofline"]["avg_launch_ms"]*1e3, max(w)))
    except Exception as e:
        print(f, "ERR", e)
PY
