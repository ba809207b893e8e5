"""Per-query work of the KNN search on the renderer's batch (build knn.hip with -DEXP_KNN_STATS: the kernel writes its
counters into the distance rows instead of the result).  tools/knn_stats.sh"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
npc, dec, ren, rays = bench.build_renderer(dev)
S = ren.N_surface
nq = 61440
z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
rq = rays["radius"][:nq].repeat_interleave(S)
for ball in (False, True):
    kw = dict(weights=(2, False, True)) if ball else {}
    out = npc.index.search(pq, 8, radius_per_query=rq, image_layout=(S, rays["W"]), **kw)
    D = out[0]
    names = ["candidates", "accepted into the pending slots", "insertion rounds (per wave)", "(z, y) rows", "shells"]
    print("ball-limited (as the renderer calls it)" if ball else "plain 8-NN")
    for k, nm in enumerate(names):
        v = D[:, k]
        print(f"   {nm:34s} mean {float(v.mean()):8.1f}   p50 {float(v.median()):7.0f}   p90 {float(v.quantile(0.9)):7.0f}   max {float(v.max()):7.0f}")
