#!/bin/bash
# HBM bytes of the dominant kernel at config 5's shape (one rank's share of 10^8 vertices / 10^9 edges, dg x 8): the eager loop
# under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only ("no PMC pass for that workload").
out=${1:-gpurun_out/profiles}; mkdir -p $out
export TMPDIR=/tmp
R=$PWD
flags="--gpus 1 --no-configs --skip-cpu-baseline --skip-opt-hit --skip-microbench --skip-reference-equivalent --no-graph --steps 60 --vertices 100000000 --edges 1000000000 --as-rank-of 8 --cache-ratio 0.30"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && PG_MISSQ_HOST_WAIT=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc5_$c -o p -- \
        python "$R/bench.py" $flags > /tmp/pmc5_$c.log 2>&1 ); tail -2 /tmp/pmc5_$c.log | cut -c1-300
done
python - "$out/pmc_config5_rank_of_8.json" <<'EOF'
import csv, json, sys, collections
def load(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "pg::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:64]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return acc
f = load("/tmp/pmc5_FETCH_SIZE/p_counter_collection.csv", "FETCH_SIZE")
w = load("/tmp/pmc5_WRITE_SIZE/p_counter_collection.csv", "WRITE_SIZE")
out = {}
for k in sorted(f):
    if k not in w:
        continue
    n = len(f[k])
    out[k] = {"launches": n, "fetch_bytes_corrected_per_launch": 2 * 1024 * sum(v for v, _ in f[k]) / n,
              "write_bytes_per_launch": 1024 * sum(v for v, _ in w[k]) / len(w[k]), "avg_ns_under_pmc": sum(d for _, d in f[k]) / n}
    out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_corrected_per_launch"] + out[k]["write_bytes_per_launch"]
json.dump({"workload": "bench.py --vertices 1e8 --edges 1e9 --as-rank-of 8 --cache-ratio 0.30 --no-graph --steps 60 (eager loop, host-side miss waits)",
           "corrections": "FETCH_SIZE x 2, WRITE_SIZE x 1, KiB (gfx950)", "kernels": out}, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    if v["launches"] >= 30 and v["hbm_bytes_per_launch"] > 1e6:
        print("%-64s n=%5d  fetch %9.3f MB  write %8.3f MB  %8.1f us" % (k, v["launches"], v["fetch_bytes_corrected_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["avg_ns_under_pmc"] / 1e3))
EOF
grep -o '"edges_per_launch": [0-9.]*\|"destinations_per_launch": [0-9.]*\|"algorithmic_bytes_per_launch": [0-9.]*' /tmp/pmc5_FETCH_SIZE.log | head -3
