#!/bin/bash
# decoder kernel variants (GLORIE_MLP_VARIANT: v4 | n2w8 | n2w4): render tests + per-kernel times of 4 full-frame passes
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; : > gpurun_out/exp_mlp.txt
for v in "$@"; do
  echo "=== GLORIE_MLP_COL_VARIANT=$v" >> gpurun_out/exp_mlp.txt
  GLORIE_MLP_COL_VARIANT=$v python -m pytest tests/test_gpu_render.py -x -q 2>&1 | tail -2 >> gpurun_out/exp_mlp.txt
  (cd /tmp && rm -rf /tmp/prof && GLORIE_MLP_COL_VARIANT=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/prof_render.py > /tmp/log 2>&1
   f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python $R/tools/show_stats.py $f 8 >> $R/gpurun_out/exp_mlp.txt)
done
cat gpurun_out/exp_mlp.txt
