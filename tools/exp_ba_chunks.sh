export GLORIE_EXTRA_HIPFLAGS_ONLY=ba.hip
for flags in "" "-DEXP_BA_CHUNK_WGS=128" "-DEXP_BA_CHUNK_WGS=64" ""; do
  echo "== flags: [$flags]"
  GLORIE_EXTRA_HIPFLAGS="$flags" python glorie_slam_amd/build.py > /dev/null 2>&1 || { echo build failed; continue; }
  python tools/time_ba_g8.py 2>&1 | grep -v amdgpu
done
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
