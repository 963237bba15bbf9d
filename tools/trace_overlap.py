"""which kernels of OTHER queues run while a given kernel runs? usage: trace_overlap.py kernel_trace.csv <substring>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ev.sort()
tgt = [e for e in ev if key in e[2]]
tgt = tgt[len(tgt) // 4:]            # steady state
import bisect
starts = [e[0] for e in ev]
acc = collections.Counter(); tot = 0
for s, e, n, q in tgt:
    tot += e - s
    i = bisect.bisect_left(starts, s - 2_000_000)
    while i < len(ev) and ev[i][0] < e:
        s2, e2, n2, q2 = ev[i]
        if q2 != q and e2 > s:
            acc[n2.split("(")[0][:60]] += min(e, e2) - max(s, s2)
        i += 1
print(f"{key}: {len(tgt)} launches, mean {tot / len(tgt) / 1e3:.1f} us; share of its run time during which another queue runs:")
for n, t in acc.most_common(15):
    print(f"  {t / tot * 100:5.1f} %  {n}")
