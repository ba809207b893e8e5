#!/bin/bash
# phase timeline of conv_pp_kernel -> gpurun_out/conv_pp_timeline$2.txt
export GLORIE_EXTRA_HIPFLAGS_ONLY=conv.hip
mkdir -p gpurun_out
GLORIE_EXTRA_HIPFLAGS="-DEXP_CONV_STAMPS $1" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
python tools/conv_pp_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_pp_timeline$2.txt
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/conv_pp_timeline$2.txt
