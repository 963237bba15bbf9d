"""print, for a few consecutive minibatches, when each pipeline stage ran (from a rocprofv3 kernel trace)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
copies = []
if len(sys.argv) > 2:
    copies = [r for r in csv.DictReader(open(sys.argv[2])) if r["Direction"].endswith("HOST_TO_DEVICE")]
packs = [r for r in rows if "k_pack" in r["Kernel_Name"]]
t0 = int(packs[-40]["Start_Timestamp"]); t1 = int(packs[-34]["Start_Timestamp"])
ev = []
prev_q5_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > t1: continue
    n = r["Kernel_Name"]; q = r["Queue_Id"]
    tag = None
    for k, name in (("k_seed_layer", "SAMPLE begin"), ("k_pack", "SAMPLE end"), ("k_split", "LOAD split"), ("k_gather", "LOAD gather"),
                    ("k_publish", "LOAD publish"), ("k_scatter<", "MISS scatter"), ("k_signal", "MISS signal"), ("k_wait_landed", "COMPUTE wait-kernel")):
        if k in n: tag = name
    if tag: ev.append((s, e, q, tag))
# compute stream = the queue with the most kernels
from collections import Counter
cq = Counter(r["Queue_Id"] for r in rows if "k_wait_landed" in r["Kernel_Name"] or "k_linear_fwd" in r["Kernel_Name"]).most_common(1)[0][0]
comp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Queue_Id"] == cq and t0 <= int(r["Start_Timestamp"]) <= t1 and "k_wait_landed" not in r["Kernel_Name"]]
# group compute kernels into bursts separated by > 20 us gaps
burst_s = None; last_e = None
for s, e in comp:
    if burst_s is None: burst_s, last_e = s, e
    elif s - last_e > 20000:
        ev.append((burst_s, last_e, cq, "COMPUTE burst")); burst_s, last_e = s, e
    else: last_e = max(last_e, e)
if burst_s is not None: ev.append((burst_s, last_e, cq, "COMPUTE burst"))
for r in copies:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s <= t1: ev.append((s, e, "sdma", "MISS h2d copy"))
for s, e, q, tag in sorted(ev):
    print(f"+{(s-t0)/1e3:9.1f} .. +{(e-t0)/1e3:9.1f} us  q{q:5s} {tag}")
