#!/bin/bash
export GLORIE_EXTRA_HIPFLAGS_ONLY=conv.hip
GLORIE_EXTRA_HIPFLAGS="-DEXP_CONV_STAMPS" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
python tools/conv_ps_timeline.py "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_ps_timeline.txt
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/conv_ps_timeline.txt
