#!/bin/bash
# why does the stand-alone two-table gather write 2.3x its rows (r02 PMC)?  WRITE_SIZE next to the L2 -> fabric write request
# counters of idw_gather2_kernel -> gpurun_out/pmc_gather_writes.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_gather_writes.txt; : > $O
for set in "WRITE_SIZE" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_ATOMIC_sum TCC_EA_WR_UNCACHED_32B_sum" "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum" "FETCH_SIZE"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/profiles/pmc_gathers.py > /tmp/log 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "idw_gather" in n or "copyBuffer" in n:
        acc[n[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n, {k: (round(sum(v) / len(v)), len(v)) for k, v in d.items()})
PY
done
cat $O
