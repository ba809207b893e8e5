#!/bin/bash
# MFMA utilisation of the convolution and decoder kernels (derived counter MfmaUtil = MFMA-busy cycles / CU-busy
# cycles, and the raw counters behind it), separate --pmc passes -> gpurun_out/pmc_mfma.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/pmc_mfma.txt
for set in "MfmaUtil" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU"; do
  for wl in "profiles/pmc_gathers.py" "tools/prof_render.py"; do
    rm -rf /tmp/pm; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm -o p -- python $R/$wl > /tmp/log 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$set" >> $R/gpurun_out/pmc_mfma.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "conv_igemm_kernel<1" in n or "conv_igemm_kernelILi1" in n or "conv_pp_kernel<1" in n or "conv_pp_kernelILi1" in n or "mlp_" in n:
        short = n.split("(")[0].replace("void ", "").replace("glorie::", "")[:34]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(acc.items()):
    print(f"{n:36s}", {k: round(sum(v) / len(v), 2) for k, v in d.items()}, f"launches {len(next(iter(d.values())))}")
PY
  done
done
cat $R/gpurun_out/pmc_mfma.txt
