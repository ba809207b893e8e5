import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from bench_knn import timed
dev = torch.device("cuda:0")
npc, dec, ren, rays = bench.build_renderer(dev, 0, 1)
S = ren.N_surface
nq = 61440
z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
rq = rays["radius"][:nq].repeat_interleave(S)
for lay in ((S, 640), None):
    ts = [timed(lambda: npc.index.search(pq, 8, radius_per_query=rq, image_layout=lay, weights=(2, False, True)), 10) for _ in range(3)]
    print("layout", lay, "product search ms", [round(t, 4) for t in ts])
