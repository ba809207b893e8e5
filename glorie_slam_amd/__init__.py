"""glorie_slam_amd -- MI355X-native (gfx950) dense hot path of GlORIE-SLAM.

The package holds only what the hot path needs:
  csrc/            hand-written HIP kernels + the C ABI (include/glorie_hip.h)
  _lib.py          ctypes loader of lib/libglorie_hip.so (no CPU fallback)
  droid_backends   drop-in for the reference's pybind module of the same name
  synth            seeded synthetic inputs of BASELINE.md (graphs, clouds, rays)
"""
__version__ = "0.1.0"
