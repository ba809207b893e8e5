"""Prints the kernels of one BA-update step of a rocprofv3 kernel trace (csv): everything between two consecutive launches
of a marker kernel.  Usage: python tools/trace_step.py <kernel_trace.csv> [marker] [k]
k omitted: the fullest window (latest wins); k given: the k-th window that holds 25..60 kernels (a regular step)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "corr_lookup"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
pairs = list(zip(idx[:-1], idx[1:]))
if len(sys.argv) > 3:
    a, b = [ab for ab in pairs if 25 <= ab[1] - ab[0] <= 60][int(sys.argv[3])]
else:
    a, b = max(pairs, key=lambda ab: (ab[1] - ab[0], ab[0]))
tot = 0.0
t_first = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f'{(int(r["Start_Timestamp"]) - t_first) / 1e3:9.1f} {d:8.1f}  {r["Kernel_Name"][:100]}')
span = (int(rows[b]["Start_Timestamp"]) - t_first) / 1e3
print(f"kernels {b - a}  busy {tot:.1f} us  span {span:.1f} us")
