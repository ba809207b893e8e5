"""mapping iteration: eager against recorded (SequenceRunner.map_keyframe) on fixed keyframes - cost of one eager iteration, of
the recording, and of a replay.   python tools/prof_map_graph.py [map_iters] [map_rays]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glorie_slam_amd.pipeline import synthetic_images, synthetic_runner  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
K = 8
for graphs in (False, True, False, True):
    run, c = synthetic_runner(dev, K, zero_flow_head=True, map_iters=M, map_rays=R)
    run.map_graph = graphs
    video, imgs = c["video"], synthetic_images(K)
    video.poses[:K] = c["poses"][:K]
    video.disps[:K] = c["disps"][:K]
    video.disps_up[:K] = torch.nn.functional.interpolate(c["disps"][:K, None], scale_factor=8, mode="bilinear",
                                                         align_corners=False)[:, 0]
    video.counter.value = K
    cap_ms = []
    real = run._capture_iteration

    def timed_capture(it):
        torch.cuda.synchronize()
        t = time.perf_counter()
        o = real(it)
        torch.cuda.synchronize()
        cap_ms.append(1e3 * (time.perf_counter() - t))
        return o
    run._capture_iteration = timed_capture
    for k in range(K):
        run.images[k] = imgs[k].to(dev)
        run.map_keyframe(k)
    per_it = run.timing["map_iter_ms"][2:]
    line = f"graphs={graphs}: {sum(per_it) / len(per_it):.3f} ms per iteration over {M} iterations ({R} rays)"
    if cap_ms:
        cap = sum(cap_ms[2:]) / len(cap_ms[2:])
        tot = sum(per_it) / len(per_it) * M
        line += f"; recording {cap:.2f} ms per keyframe; (total - recording) / {M} = {(tot - cap) / M:.3f} ms"
    print(line, flush=True)
