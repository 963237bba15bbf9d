mkdir -p gpurun_out/r03/share
SK="--skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
for sh in 0.5 0.3125 0.0; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --host-threads 2 --cpu-share $sh $SK > gpurun_out/r03/share/t2_s$sh.json 2> gpurun_out/r03/share/t2_s$sh.log
done
python bench.py --gpus 1 --steps 20 --warmup 5 $SK > gpurun_out/r03/share/t12_auto.json 2> gpurun_out/r03/share/t12_auto.log
taskset -c 0-3 python bench.py --gpus 1 --steps 20 --warmup 5 $SK > gpurun_out/r03/share/taskset4_auto.json 2> gpurun_out/r03/share/taskset4_auto.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/share/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; w=d["ms_per_step_windows"]
        print(f.split("/")[-1], "epoch ms/step %.4f win %.4f share %s threads %s gather_us %.0f rows_pcie %.0f fused_us %.1f maxwin %.3f" % (c["epoch_ms_per_step"], d["ms_per_step"], c["cpu_share"], d["host"]["miss_gather_threads"], d["miss_queue"]["timed_region"]["us_cpu_gather"], d["miss_queue"]["timed_region"]["rows_over_pcie_per_step"], d["roofline"]["avg_launch_ms"]*1e3, max(w)))
    except Exception as e:
        print(f, "ERR", e)
PY
PG_MISSQ_TAIL_STREAM=caller python bench.py --gpus 1 --steps 20 --warmup 5 --host-threads 2 --cpu-share 0.3125 $SK > gpurun_out/r03/share/t2_s0.3125_callerstream.json 2> gpurun_out/r03/share/t2_callerstream.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/share/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; w=d["ms_per_step_windows"]
        print(f.split("/")[-1], "epoch ms/step %.4f win %.4f share %s threads %s gather_us %.0f rows_pcie %.0f fused_us %.1f maxwin %.3f" % (c["epoch_ms_per_step"], d["ms_per_step"], c["cpu_share"], d["host"]["miss_gather_threads"], d["miss_queue"]["timed_region"]["us_cpu_gather"], d["miss_queue"]["timed_region"]["rows_over_pcie_per_step"], d["roofline"]["avg_launch_ms"]*1e3, max(w)))
    except Exception as e:
        print(f, "ERR", e)
PY
