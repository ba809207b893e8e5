#!/bin/bash
# kernel trace of a short bench run, then the kernels of one regular BA-update step (through gpurun):
#   tools/step_trace.sh [k]   -> gpurun_out/step_trace.txt
R=$PWD; K=${1:-20}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st; rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o s -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/st.log 2>&1
mkdir -p $R/gpurun_out
python $R/tools/trace_step.py $(find /tmp/st -name "*kernel_trace.csv" | head -1) corr_lookup $K > $R/gpurun_out/step_trace.txt 2>&1
cut -c1-140 $R/gpurun_out/step_trace.txt
