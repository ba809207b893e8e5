"""Prints the top rows of a rocprofv3 kernel_stats.csv: name, calls, average us, percentage."""
import csv
import sys

for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(f"{r['Name'][:48]:48s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:10.1f} us {r['Percentage']:>6s}%")
