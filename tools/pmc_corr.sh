#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the correlation lookups in separate --pmc passes -> gpurun_out/pmc_corr/{fetch,write}.csv + log
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pc_fetch /tmp/pc_write; mkdir -p $R/gpurun_out/pmc_corr
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pc_fetch -o g -- python $R/tools/pmc_corr.py > $R/gpurun_out/pmc_corr/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pc_write -o g -- python $R/tools/pmc_corr.py > $R/gpurun_out/pmc_corr/write.log 2>&1
cp $(find /tmp/pc_fetch -name "g_counter_collection.csv" | head -1) $R/gpurun_out/pmc_corr/fetch.csv
cp $(find /tmp/pc_write -name "g_counter_collection.csv" | head -1) $R/gpurun_out/pmc_corr/write.csv
python $R/tools/pmc_corr_summary.py $R/gpurun_out/pmc_corr | tee $R/gpurun_out/pmc_corr/summary.json
