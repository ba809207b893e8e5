#!/bin/bash
# shipped convolution kernels with experiment flags     usage: exp_conv.sh "<flags A>" "<flags B>" ...
export GLORIE_EXTRA_HIPFLAGS_ONLY=conv.hip
mkdir -p gpurun_out; : > gpurun_out/exp_conv.txt
for fl in "$@"; do
  GLORIE_EXTRA_HIPFLAGS="$fl" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
  echo "== flags: [$fl]" >> gpurun_out/exp_conv.txt
  python tools/bench_conv.py 2>&1 | grep "3x3\|gate 320\|q gate\|heads" | sed 's/miopen.*igemm/igemm/' >> gpurun_out/exp_conv.txt
done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/exp_conv.txt
