#!/bin/bash
# KNN search with experiment flags     usage: exp_knn.sh "<flags A>" "<flags B>" ...
export GLORIE_EXTRA_HIPFLAGS_ONLY=knn.hip
mkdir -p gpurun_out; : > gpurun_out/exp_knn.txt
for fl in "$@"; do
  GLORIE_EXTRA_HIPFLAGS="$fl" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
  echo "== flags: [$fl]" >> gpurun_out/exp_knn.txt
  python tools/bench_knn.py 2>&1 | grep "^rays" >> gpurun_out/exp_knn.txt
done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/exp_knn.txt
