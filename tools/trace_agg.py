"""Aggregates the kernels of the last steady-state BA-update steps of a rocprofv3 kernel trace (csv):
one block per step type.  Usage: python tools/trace_agg.py <kernel_trace.csv> [min_us]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "reproject_kernel" in r["Kernel_Name"]]
wins = list(zip(idx[:-1], idx[1:]))
for a, b in wins[-4:-2]:
    t0 = int(rows[a]["Start_Timestamp"])
    agg = collections.OrderedDict()
    tot = 0.0
    for r in rows[a:b]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        k = r["Kernel_Name"][:70]
        c, t = agg.get(k, (0, 0.0))
        agg[k] = (c + 1, t + d)
    print("---- step: kernels", b - a, "busy", round(tot, 1), "span", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
    small, ns = 0.0, 0
    for k, (c, t) in agg.items():
        if t >= thr:
            print(f"{c:3d} {t:8.1f}  {k}")
        else:
            small += t
            ns += c
    print(f"{ns:3d} {small:8.1f}  (kernels below {thr} us in total)")
