#!/bin/bash
# conv_pp_kernel experiments: one build per flag set, interleaved A/B of the z|r launch against the default tile + phase timeline
export GLORIE_EXTRA_HIPFLAGS_ONLY=conv.hip
for flags in "$@"; do
  echo "== flags: [$flags]"
  GLORIE_EXTRA_HIPFLAGS="$flags" python glorie_slam_amd/build.py > /dev/null 2>&1 || { echo build failed; continue; }
  python -c "
import sys, os; sys.path.insert(0, 'tools'); os.environ['BENCH_CONV_GATE'] = '0'
import bench_conv; bench_conv.pp_ab(4)" 2>&1 | grep -v amdgpu.ids
done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
