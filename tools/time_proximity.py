"""Wall-clock of FactorGraph.add_backend_proximity_factors (frame_distance kernel + edge selection with
suppression, factor_graph.py:386-462 of the reference) at 128 / 512 keyframes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glorie_slam_amd.synth as synth
from glorie_slam_amd.depth_video import DepthVideo
from glorie_slam_amd.droid_net import UpdateModule
from glorie_slam_amd.factor_graph import FactorGraph
dev="cuda:0"; h,w=30,40
net = UpdateModule().to(dev).eval()
for K in (128, 512):
    cfg = {"cam": {"H_out": 8*h, "W_out": 8*w}, "device": dev, "setting":"t","scene":"s","data":{"output":"/tmp"},
           "tracking": {"buffer": K+8, "beta":0.75,"warmup":8,"max_age":50,"mono_thres":0.1,
                        "multiview_filter":{"thresh":0.25,"visible_num":2},"store_images":False,
                        "backend":{"BA_type":"DSPO"}}}
    g = synth.loop_graph(K=K, h=h, w=w)
    video = DepthVideo(cfg)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    video.poses[:K]=t(g["poses"][:K]); video.disps[:K]=t(g["disps"][:K]); video.intrinsics[:]=t(g["intrinsics"][0])
    video.counter.value=K
    for rep in range(2):
        graph = FactorGraph(video, net, device=dev, corr_impl='alt', max_factors=6*K)
        torch.cuda.synchronize(); t0=time.perf_counter()
        n = graph.add_backend_proximity_factors(0, K, 5, 1, 25.0, 6*K, 0.75)
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"K={K}: add_backend_proximity_factors {dt*1e3:.1f} ms -> {n} edges", flush=True)
