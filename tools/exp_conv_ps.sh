#!/bin/bash
# producer / consumer convolution: timings with ablation flags    usage: exp_conv_ps.sh "<flags A>" "<flags B>" ...
export GLORIE_EXTRA_HIPFLAGS_ONLY=conv.hip GLORIE_CONV_PS=1
mkdir -p gpurun_out; : > gpurun_out/exp_conv_ps.txt
for fl in "$@"; do
  GLORIE_EXTRA_HIPFLAGS="$fl" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
  echo "== flags: [$fl]" >> gpurun_out/exp_conv_ps.txt
  python tools/bench_conv.py 2>&1 | grep "3x3\|1x1" | sed 's/miopen.*igemm/igemm/' >> gpurun_out/exp_conv_ps.txt
done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/exp_conv_ps.txt
