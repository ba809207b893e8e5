"""Wall-clock of Frontend() per keyframe on a synthetic 640x480 sequence (60x80 BA resolution): bootstrap
(16 pose_depth iterations on 8 keyframes) and steady-state keyframes (8 + 4 alternating DSPO iterations,
edge management, redundancy test, valid-depth masks).  Default-init update operator, so the trajectory is
meaningless - only the cost of the control flow + kernels is."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glorie_slam_amd.synth as synth  # noqa: E402
from glorie_slam_amd.depth_video import DepthVideo  # noqa: E402
from glorie_slam_amd.droid_net import UpdateModule  # noqa: E402
from glorie_slam_amd.frontend import Frontend  # noqa: E402

dev = "cuda:0"
h, w, K = 60, 80, 24
cfg = {
    "cam": {"H_out": 8 * h, "W_out": 8 * w}, "device": dev, "setting": "t", "scene": "s", "data": {"output": "/tmp"},
    "tracking": {"buffer": 32, "beta": 0.75, "warmup": 8, "max_age": 50, "mono_thres": 0.1,
                 "multiview_filter": {"thresh": 0.25, "visible_num": 2}, "store_images": False,
                 "frontend": {"enable_loop": False, "keyframe_thresh": 0.0, "thresh": 16.0, "window": 25,
                              "radius": 1, "nms": 1, "max_factors": 75},
                 "backend": {"BA_type": "DSPO", "thresh": 25.0, "radius": 1, "nms": 5, "normalize": False,
                             "loop_window": 25, "loop_thresh": 25.0, "loop_radius": 1, "loop_nms": 12}}}
g = synth.keyframe_graph(K=K, h=h, w=w, radius=3)
fmaps, nets, inps = synth.feature_maps(K, h, w)
video = DepthVideo(cfg)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
video.poses[:K] = t(g["poses"][:K]); video.disps[:K] = t(g["disps"][:K]); video.intrinsics[:] = t(g["intrinsics"][0])
video.fmaps[:K] = t(fmaps); video.nets[:K] = t(nets); video.inps[:K] = t(inps)
video.mono_disps[:K] = t(g["disps"][:K] * 0.8 + 0.01)
torch.manual_seed(43)
net = types.SimpleNamespace(update=UpdateModule().to(dev).eval())
for use_graphs in (False, True):
    video.poses[:K] = t(g["poses"][:K]); video.disps[:K] = t(g["disps"][:K])
    video.counter.value = 8
    fe = Frontend(net, video, cfg, use_graphs=use_graphs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fe()
    torch.cuda.synchronize(); boot = time.perf_counter() - t0
    per = []
    for k in range(9, K):
        video.counter.value = k
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fe()
        torch.cuda.synchronize(); per.append(time.perf_counter() - t0)
    per = np.array(per) * 1e3
    print(f"use_graphs={use_graphs}: bootstrap {boot * 1e3:.1f} ms; keyframes: median {np.median(per):.1f} ms, "
          f"last {per[-1]:.1f} ms, edges {fe.graph.ii.numel()} (+{fe.graph.ii_inac.numel()} inactive)", flush=True)
