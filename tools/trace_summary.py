"""per-queue / per-kernel busy time from a rocprofv3 kernel_trace.csv, restricted to the last `frac` of the run"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows)
# timed region = after the last k_rmat/k_features... take the window of the last N k_pack launches
packs = [r for r in rows if "k_pack" in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
w0 = int(packs[-n]["Start_Timestamp"]); w1 = int(packs[-1]["End_Timestamp"])
sel = [r for r in rows if w0 <= int(r["Start_Timestamp"]) <= w1]
print(f"window {(w1-w0)/1e6:.2f} ms, {n-1} steps -> {(w1-w0)/1e3/(n-1):.1f} us/step, {len(sel)} kernels ({len(sel)/(n-1):.1f}/step)")
byq = collections.defaultdict(float); byk = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    byq[r.get("Queue_Id", "?")] += d
    k = r["Kernel_Name"][:70]
    byk[k][0] += 1; byk[k][1] += d
for q, d in sorted(byq.items(), key=lambda x: -x[1]):
    print(f"queue {q}: busy {d/1e3/(n-1):8.1f} us/step")
for k, (c, d) in sorted(byk.items(), key=lambda x: -x[1][1])[:28]:
    print(f"{d/1e3/(n-1):8.1f} us/step  {c/(n-1):5.2f}/step  avg {d/c/1e3:7.1f} us  {k}")
# ---- concurrency: union of busy intervals vs per-queue sums
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
u = 0; cs, ce = iv[0]
for a, b in iv[1:]:
    if a > ce:
        u += ce - cs; cs, ce = a, b
    else:
        ce = max(ce, b)
u += ce - cs
print(f"union busy {u/1e3/(n-1):.1f} us/step  (sum of queues {sum(byq.values())/1e3/(n-1):.1f}) idle {(w1-w0-u)/1e3/(n-1):.1f} us/step")
# timeline of one step in the middle: list kernels with queue, start offset, duration
mid = packs[-n // 2]
t0 = int(mid["Start_Timestamp"])
one = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t0 + 1_400_000]
for r in one[:140]:
    print(f"q{r['Queue_Id']} +{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  {r['Kernel_Name'][:60]}")
