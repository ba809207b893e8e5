#!/bin/bash
# ablations of the decoder kernels (csrc/mlp.hip): per-kernel times of 4 full-frame passes per flag set
# usage: exp_mlp_ablate.sh "<flags A>" "<flags B>" ...  -> gpurun_out/exp_mlp_ablate.txt
R=$PWD; export TMPDIR=/tmp; export GLORIE_EXTRA_HIPFLAGS_ONLY=mlp.hip
mkdir -p gpurun_out; : > gpurun_out/exp_mlp_ablate.txt
if [ $# -eq 0 ]; then set -- "" "-DEXP_MLP_NO_SOFTPLUS" "-DEXP_MLP_NO_SPLIT" "-DEXP_MLP_NO_SOFTPLUS -DEXP_MLP_NO_SPLIT"; fi
for fl in "$@"; do
  GLORIE_EXTRA_HIPFLAGS="$fl" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
  echo "== flags: [$fl]" >> gpurun_out/exp_mlp_ablate.txt
  (cd /tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/prof_render.py > /tmp/log 2>&1
   f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python $R/tools/show_stats.py $f 5 | grep "nb_v\|col_v" >> $R/gpurun_out/exp_mlp_ablate.txt)
done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/exp_mlp_ablate.txt
