#!/bin/bash
# wave timeline of the fused displacement-major lookup -> gpurun_out/exp_corr_timeline.txt
export GLORIE_EXTRA_HIPFLAGS_ONLY=corr_dm.hip
mkdir -p gpurun_out
GLORIE_EXTRA_HIPFLAGS="-DEXP_DM_TIMESTAMPS" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
python tools/exp_corr_timeline.py "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/exp_corr_timeline.txt
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/exp_corr_timeline.txt
