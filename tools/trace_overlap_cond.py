"""Does a kernel run longer when kernel X of another queue runs beside it?  For every launch of <substring> (steady state)
the share of its run time overlapped by each other kernel name; then, per other kernel: mean duration of the launches it
overlaps for > 30 % of their time vs of those it does not touch (< 1 %).
usage: trace_overlap_cond.py kernel_trace.csv <substring>"""
import bisect, collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70], r["Queue_Id"]) for r in rows)
tgt = [e for e in ev if key in e[2]]
tgt = tgt[len(tgt) // 4:]
starts = [e[0] for e in ev]
per = []                      # (duration, {name: overlapped share})
for s, e, n, q in tgt:
    ov = collections.Counter()
    i = bisect.bisect_left(starts, s - 2_000_000)
    while i < len(ev) and ev[i][0] < e:
        s2, e2, n2, q2 = ev[i]
        if q2 != q and e2 > s:
            ov[n2] += (min(e, e2) - max(s, s2)) / (e - s)
        i += 1
    per.append((e - s, ov))
names = collections.Counter()
for d, ov in per:
    for n in ov:
        names[n] += 1
alone = [d for d, ov in per if sum(ov.values()) < 0.01]
print(f"{key}: {len(per)} launches, mean {sum(d for d, _ in per) / len(per) / 1e3:.1f} us; "
      f"{len(alone)} with no other queue busy: mean {sum(alone) / max(len(alone), 1) / 1e3:.1f} us")
for n, _ in names.most_common(12):
    hi = [d for d, ov in per if ov.get(n, 0) > 0.30]
    lo = [d for d, ov in per if ov.get(n, 0) < 0.01]
    if hi and lo:
        print(f"  {n:70s} overlapped > 30 %: {len(hi):4d} launches {sum(hi) / len(hi) / 1e3:6.1f} us | untouched: {len(lo):4d} launches {sum(lo) / len(lo) / 1e3:6.1f} us")
