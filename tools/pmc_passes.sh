#!/bin/bash
# HBM-traffic passes (separate --pmc runs, as MI355X_MICROARCH.md prescribes) -> gpurun_out/pmc_kernels.json
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_fetch /tmp/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -o g -- python $R/profiles/pmc_gathers.py > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -o g -- python $R/profiles/pmc_gathers.py > /tmp/pw.log 2>&1
fd=$(dirname $(find /tmp/pmc_fetch -name "g_counter_collection.csv" | head -1)); wd=$(dirname $(find /tmp/pmc_write -name "g_counter_collection.csv" | head -1))
python $R/profiles/summarize_pmc.py $fd $wd > $R/gpurun_out/pmc_kernels.json
head -c 1500 $R/gpurun_out/pmc_kernels.json
