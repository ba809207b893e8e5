#!/bin/bash
# PMC passes over the KNN query / gather kernels of the bench workload (profiles/pmc_gathers.py)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD TA_BUSY_avr TA_TA_BUSY_sum"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/profiles/pmc_gathers.py > /tmp/log 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "knn_query" in n or "idw_gather" in n:
        acc[n[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n, {k: round(sum(v) / len(v)) for k, v in d.items()})
PY
done
