#!/bin/bash
# the CPU row gather of the miss path under 2 / 4 / 8 threads, with and without pinned gather threads
# (PG_MISSQ_PIN=2) and a huge-page backed host table (PG_HOST_TABLE_THP=1): TEN epochs per point (10 840 steps, 542 windows of
# 20 steps), window min / p10 / p50 / p90 / max, CPUs used by the process, CPU-gather time per job.
# usage: tools/exp_host_sweep.sh <out dir>
OUT=${1:-gpurun_out/host_sweep}; mkdir -p "$OUT"
S="--skip-microbench --skip-cpu-baseline --skip-opt-hit --skip-reference-equivalent --no-configs --steps 10840 --warmup 20 --no-adapt-cpu-share"
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null | sed 's/^/# THP: /'
for t in 2 4 8; do for v in base pin thp pinthp; do
  E=""; case $v in pin) E="PG_MISSQ_PIN=2";; thp) E="PG_HOST_TABLE_THP=1";; pinthp) E="PG_MISSQ_PIN=2 PG_HOST_TABLE_THP=1";; esac
  env $E timeout 300 python bench.py $S --host-threads $t > "$OUT/bench_t${t}_$v.json" 2> "$OUT/bench_t${t}_$v.err"
done; done
python - "$OUT" <<'PY'
import json, sys, glob, os
print(f"{'point':14s} {'ms/step':>8s} {'min':>7s} {'p10':>7s} {'p50':>7s} {'p90':>7s} {'max':>7s} {'cpus':>5s} {'gather_us':>9s} {'windows>0.2':>11s}")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_t*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); q = d["ms_per_step_window_quantiles"]; w = d["ms_per_step_windows"]
        tr = (d["miss_queue"] or {}).get("timed_region") or {}
        print(f"{os.path.basename(f)[6:-5]:14s} {d['ms_per_step']:8.4f} {q['min']:7.4f} {q['p10']:7.4f} {q['p50']:7.4f} {q['p90']:7.4f} {q['max']:7.4f} "
              f"{d['host']['timed_region_cgroup'].get('process_cpus_used', 0):5.2f} {tr.get('us_cpu_gather', 0):9.1f} {sum(1 for x in w if x > 0.2):11d}")
    except Exception as e:
        print(os.path.basename(f), "unreadable", e, open(f[:-5] + ".err").read()[-300:])
PY
