#!/bin/bash
# usage: exp_variants.sh "<flags A>" "<flags B>" ...  : rebuild with each flag set and print the render kernel stats
R=$PWD
export TMPDIR=/tmp
for fl in "$@"; do
  echo "=== variant: [$fl]"
  GLORIE_EXTRA_HIPFLAGS="$fl" python glorie_slam_amd/build.py || exit 1
  (cd /tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/prof_render.py > /tmp/log 2>&1
   f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python $R/tools/show_stats.py $f 5 | grep -E "mlp|knn")
done
