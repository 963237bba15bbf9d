"""kernel sequence of one replayed step on the busiest queue, durations averaged over the steps that have
the modal kernel count.  usage: trace_seq.py <rocprofv3 kernel_trace.csv>"""
import csv, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cq = Counter(r["Queue_Id"] for r in rows if "k_wait_landed" in r["Kernel_Name"] or "k_linear_fwd" in r["Kernel_Name"]).most_common(1)[0][0]
q = [r for r in rows if r["Queue_Id"] == cq]
# a step ends with the optimiser kernel (the wait kernel at its start is skipped when the miss rows are already
# on their way: pg_missq_wait_device then uses an event)
end = "k_adam" if any("k_adam" in r["Kernel_Name"] for r in q) else "multi_tensor_apply"
idx = [i for i, r in enumerate(q) if end in r["Kernel_Name"] and "FusedOptimizer" in r["Kernel_Name"] or "k_adam" in r["Kernel_Name"]]
steps = [q[a + 1:b + 1] for a, b in zip(idx[:-1], idx[1:])]
L = Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == L][5:]
tot = 0.0
for pos in range(L):
    n = steps[0][pos]["Kernel_Name"]
    d = sum(int(s[pos]["End_Timestamp"]) - int(s[pos]["Start_Timestamp"]) for s in steps) / len(steps) / 1e3
    tot += d
    for k in ("FillFunctor", "fused_dropout", "multi_tensor_apply", "nll_loss", "softmax", "Cijk", "masked_scale", "elementwise_kernel", "reduce_kernel", "CatArray"):
        if k in n: n = k + " :: " + n[n.find("<"):][:60]; break
    print(f"{d:7.1f} us  {n[:110]}")
span = sum(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in steps) / len(steps) / 1e3
print(f"sum {tot:.1f} us, first-start..last-end {span:.1f} us, over {len(steps)} steps of {L} kernels")
# step-to-step: period (first kernel start to the next step's first kernel start) and the gap between a step's last kernel
# and the next step's first one — what the stream spends between two graph replays (event records, waits, launch)
per, gap = [], []
for a, b in zip(idx[5:-2], idx[6:-1]):
    s0, s1 = q[a + 1:b + 1], q[b + 1:]
    if len(s0) != L or not s1:
        continue
    per.append(int(s1[0]["Start_Timestamp"]) - int(s0[0]["Start_Timestamp"]))
    gap.append(int(s1[0]["Start_Timestamp"]) - int(s0[-1]["End_Timestamp"]))
if per:
    per.sort(); gap.sort()
    m = len(per) // 2
    print(f"step period median {per[m] / 1e3:.1f} us (p10 {per[len(per) // 10] / 1e3:.1f}, p90 {per[9 * len(per) // 10] / 1e3:.1f}); "
          f"gap between steps median {gap[m] / 1e3:.1f} us (p10 {gap[len(gap) // 10] / 1e3:.1f}, p90 {gap[9 * len(gap) // 10] / 1e3:.1f})")
