"""Turn a rocprofv3 (ROCm 7.x, rocpd sqlite) result into the per-kernel stats table that
`--stats` reports: name, calls, total/avg duration (us), percentage.

    python profiles/summarize_rocpd.py gpurun_out/prof1/prof1_results.db > profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(path, out=sys.stdout):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])


if __name__ == "__main__":
    main(sys.argv[1])
