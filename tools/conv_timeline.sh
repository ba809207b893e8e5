#!/bin/bash
# K-tile timeline of conv_igemm_kernel -> gpurun_out/conv_timeline.txt     usage: conv_timeline.sh [zr|q]
export GLORIE_EXTRA_HIPFLAGS_ONLY=conv.hip
mkdir -p gpurun_out
GLORIE_EXTRA_HIPFLAGS="-DEXP_CONV_STAMPS" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
: > gpurun_out/conv_timeline.txt
for k in ${@:-zr q}; do python tools/conv_timeline.py $k 2>&1 | grep -v amdgpu.ids >> gpurun_out/conv_timeline.txt; done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/conv_timeline.txt
