#!/bin/bash
# kernel time of a 1/8 frame shard against the wall time of the pass -> gpurun_out/render_shard_kernels.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sh8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sh8 -o s -- python $R/tools/render_shard_kernels.py 2>&1 | grep world > $R/gpurun_out/render_shard_kernels.txt
python - >> $R/gpurun_out/render_shard_kernels.txt <<PY
import csv, glob
f = glob.glob("/tmp/sh8/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    calls, t = int(r["Calls"]), float(r["TotalDurationNs"])
    if calls % 23 == 0 and calls:
        print("%-90s %5d calls %9.1f us" % (r["Name"][:90], calls, float(r["AverageNs"]) / 1e3))
        tot += t / 23
print("kernel time per pass: %.1f us" % (tot / 1e3))
PY
cat $R/gpurun_out/render_shard_kernels.txt
