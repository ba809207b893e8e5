#!/bin/bash
export GLORIE_EXTRA_HIPFLAGS_ONLY=knn.hip
GLORIE_EXTRA_HIPFLAGS="-DEXP_KNN_STATS" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
python tools/knn_stats.py 2>&1 | grep -v amdgpu.ids > gpurun_out/knn_stats.txt
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/knn_stats.txt
