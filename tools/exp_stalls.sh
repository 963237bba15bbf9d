# N whole-epoch runs: where do the launch thread's longest iterations sit? usage: exp_stalls.sh <tag> <runs> [env ...]
tag=$1; n=$2; shift 2
mkdir -p gpurun_out/r03/stalls
SK="--skip-cpu-baseline --skip-microbench --skip-opt-hit --skip-reference-equivalent"
for i in $(seq 1 $n); do env PG_TRACE_LAUNCH=1 "$@" python bench.py --gpus 1 --warmup 20 $SK > gpurun_out/r03/stalls/${tag}_$i.json 2> gpurun_out/r03/stalls/${tag}_$i.log; done
python - $tag <<'PY'
import json,glob,sys
for f in sorted(glob.glob(f"gpurun_out/r03/stalls/{sys.argv[1]}_*.json")):
    try:
        d=json.load(open(f)); w=d["ms_per_step_windows"]
        print(f.split("/")[-1], "ms/step %.4f epoch %.4f s | windows max %.3f n>0.2: %d | host longest %s | split %s | rescued %s gather_us %.0f thr %s" % (d["ms_per_step"], d["value"], max(w), sum(1 for x in w if x>0.2), d["host_longest_iterations_ms"], d["launch_thread_longest_split_ms"], d["miss_queue"]["timed_region"]["rescued_chunks"], d["miss_queue"]["timed_region"]["us_cpu_gather"], d["host"]["timed_region_cgroup"]["nr_throttled"]))
    except Exception as e:
        print(f, "ERR", e)
PY
