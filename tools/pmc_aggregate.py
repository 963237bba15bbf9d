"""Aggregate tools/pmc_summarize.py's per-(kernel, grid) rows by kernel name (launch-weighted means) and write the
per-kernel table + the in-loop record bench.py reads for `roofline.traffic`.
usage: pmc_aggregate.py pmc_bench_per_kernel_raw.json out_dir"""
import collections, json, os, sys
raw = json.load(open(sys.argv[1]))
out_dir = sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for key, v in raw.items():
    name = key.rsplit(" grid=", 1)[0]
    grid = int(key.rsplit(" grid=", 1)[1])
    if "copyBuffer" in name:
        name += " (>=64K threads)" if grid >= 65536 else " (small)"
    if "k_gather" in name or "k_split" in name:
        name += " [micro-benchmark shape]" if grid >= 1 << 20 else " [in-loop]"
    a = acc[name]
    n = v["launches"]
    a[0] += n; a[1] += n * v["fetch_bytes_corrected"]; a[2] += n * v["write_bytes"]; a[3] += n * v["avg_ns_under_pmc"]
table = {k: {"launches": a[0], "fetch_bytes_corrected_per_launch": a[1] / a[0], "write_bytes_per_launch": a[2] / a[0],
             "hbm_bytes_per_launch": (a[1] + a[2]) / a[0], "avg_ns_under_pmc": a[3] / a[0]} for k, a in sorted(acc.items())}
json.dump(table, open(os.path.join(out_dir, "pmc_bench_per_kernel.json"), "w"), indent=1)
for k, v in table.items():
    print(f"{k[:70]:70s} n={v['launches']:5d}  fetch {v['fetch_bytes_corrected_per_launch']/1e6:9.3f} MB  write {v['write_bytes_per_launch']/1e6:9.3f} MB  "
          f"{v['avg_ns_under_pmc']/1e3:8.1f} us")
for k, v in table.items():
    if "k_spmm_fwd_rows" in k:
        short = "spmm_fwd_rows"
        rec = {"kernel": k.split("(")[0] + " in-loop (eager loop, layer 0 aggregated straight from the cache + staged miss rows)",
               "launches": v["launches"], "fetch_bytes_corrected_per_launch": v["fetch_bytes_corrected_per_launch"],
               "write_bytes_per_launch": v["write_bytes_per_launch"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
               "avg_ns_under_pmc": v["avg_ns_under_pmc"],
               "collected": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, separate passes (FETCH_SIZE x2 on gfx950, "
                            "WRITE_SIZE x1, KiB; calibrated in the same run on the 2.5 GB cache-fill copies: x1.000 / x1.000), over "
                            "PG_MISSQ_HOST_WAIT=1 python bench.py --steps 60 --no-graph ... (tools/run_profiles.sh)"}
        json.dump(rec, open(os.path.join(out_dir, f"pmc_{short}_inloop.json"), "w"), indent=1)
