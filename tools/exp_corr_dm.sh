#!/bin/bash
# ablations of the fused displacement-major lookup (csrc/corr_dm.hip): full kernel, without the gathers, without the MFMAs
# usage: exp_corr_dm.sh "<flags A>" "<flags B>" ...   (default: the gather / MFMA ablations) -> gpurun_out/exp_corr_dm.txt
export GLORIE_EXTRA_HIPFLAGS_ONLY=corr_dm.hip
mkdir -p gpurun_out; : > gpurun_out/exp_corr_dm.txt
if [ $# -eq 0 ]; then set -- "" "-DEXP_DM_NO_GATHER" "-DEXP_DM_NO_MFMA" "-DEXP_DM_NO_GATHER -DEXP_DM_NO_MFMA"; fi
for fl in "$@"; do
  GLORIE_EXTRA_HIPFLAGS="$fl" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
  echo "== flags: [$fl]" >> gpurun_out/exp_corr_dm.txt
  CORR_ONLY_VOLUME=1 python tools/bench_corr.py 2>&1 | grep "displacement-major" >> gpurun_out/exp_corr_dm.txt
done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/exp_corr_dm.txt
