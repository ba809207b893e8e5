"""Wall-clock of Backend.dense_ba (global BA, on-the-fly correlation, update_lowmem) on synthetic loop
trajectories at the BASELINE config-4 resolution (240x320 -> 30x40)."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glorie_slam_amd.synth as synth  # noqa: E402
from glorie_slam_amd.backend import Backend  # noqa: E402
from glorie_slam_amd.depth_video import DepthVideo  # noqa: E402
from glorie_slam_amd.droid_net import UpdateModule  # noqa: E402

dev = "cuda:0"
h, w = 30, 40
torch.manual_seed(43)
net = types.SimpleNamespace(update=UpdateModule().to(dev).eval())
for K in [int(a) for a in sys.argv[1:]] or [32, 64, 128]:
    cfg = {"cam": {"H_out": 8 * h, "W_out": 8 * w}, "device": dev, "setting": "t", "scene": "s", "data": {"output": "/tmp"},
           "tracking": {"buffer": K + 8, "beta": 0.75, "warmup": 8, "max_age": 50, "mono_thres": 0.1,
                        "multiview_filter": {"thresh": 0.25, "visible_num": 2}, "store_images": False,
                        "frontend": {"enable_loop": False, "keyframe_thresh": 0.0, "thresh": 16.0, "window": 25,
                                     "radius": 1, "nms": 1, "max_factors": 75},
                        "backend": {"BA_type": "DSPO", "thresh": 25.0, "radius": 1, "nms": 5, "normalize": False,
                                    "loop_window": 25, "loop_thresh": 25.0, "loop_radius": 1, "loop_nms": 12}}}
    g = synth.loop_graph(K=K, h=h, w=w)
    fmaps, nets, inps = synth.feature_maps(K, h, w)
    video = DepthVideo(cfg)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    video.poses[:K] = t(g["poses"][:K]); video.disps[:K] = t(g["disps"][:K]); video.intrinsics[:] = t(g["intrinsics"][0])
    video.fmaps[:K] = t(fmaps); video.nets[:K] = t(nets); video.inps[:K] = t(inps)
    video.mono_disps[:K] = t(g["disps"][:K] * 0.8 + 0.01)
    video.counter.value = K
    be = Backend(net, video, cfg)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n, ne = be.dense_ba(steps=6)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"K={K}: dense_ba(steps=6) {dt * 1e3:.1f} ms, {ne} edges -> {dt * 1e3 / 6:.1f} ms per update_lowmem step", flush=True)
