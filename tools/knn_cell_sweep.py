"""knn_query time on the bench workload (65,536 rays x 10 samples, 524k-point cloud) as a function of the
grid cell size (results are exact for every cell size; this only moves work between cells and candidates)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from glorie_slam_amd.point_ops import KnnIndex  # noqa: E402

dev = "cuda:0"
npc, dec, ren, rays = bench.build_renderer(dev)
S, nq = 10, 65536
z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
rq = rays["radius"][:nq].repeat_interleave(S)
ref = None
for cell in [float(a) for a in sys.argv[1:]] or [0.04, 0.05, 0.06, 0.08, 0.10, 0.12]:
    idx = KnnIndex(dev, cell_size=cell, max_cells=1 << 21)
    idx.set_points(npc.cloud_pos())
    D, I, nn = idx.search(pq, 8, radius_per_query=rq)
    if ref is None:
        ref = (D.clone(), I.clone())
    same = bool(torch.equal(D, ref[0]) and torch.equal(I, ref[1]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        idx.search(pq, 8, radius_per_query=rq)
    e1.record(); torch.cuda.synchronize()
    print(f"cell {cell:.3f}  search {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us  identical={same}", flush=True)
