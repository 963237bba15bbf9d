"""timeline of a few steady-state steps from a rocprofv3 kernel trace: every dispatch with its queue, start (us, relative),
duration, name. usage: trace_window.py <kernel_trace.csv> [first_step_from_end=300] [steps=4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
heads = [i for i, r in enumerate(rows) if "k_gcn_head" in r["Kernel_Name"] or "k_sage_head" in r["Kernel_Name"]]
a, b = heads[-back], heads[-back + n]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
for r in rows[a - 12:b]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("void ", "").replace("pg::", "")
    print(f"q{q} {'    ' * q}{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:6.1f}  {nm[:48]}")
