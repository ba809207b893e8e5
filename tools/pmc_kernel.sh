#!/bin/bash
# PMC passes (own runs, kernel-trace only) over the kernels whose name contains $1, running the python command $2...
#   tools/pmc_kernel.sh corr_otf8 tools/bench_corr.py
R=$PWD; PAT=$1; shift; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/"$@" > /tmp/log 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python - "$f" "$PAT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sys.argv[2] in n:
        acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n, {k: round(sum(v) / len(v)) for k, v in d.items()})
PY
done
