"""decoder kernels alone (as bench.py's roofline_mlp times them) and a full 640x480 frame: python tools/time_mlp.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glorie_slam_amd import point_ops  # noqa: E402

dev = torch.device("cuda:0")
npc, dec, ren, rays = bench.build_renderer(dev)
S = ren.N_surface
nq = 61440
z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
rq = rays["radius"][:nq].repeat_interleave(S)
D_, I_, nn_ = npc.index.search(pq, 8, radius_per_query=rq)
cg_, has_, w_ = point_ops.idw_gather(D_, I_, nn_, npc.geo_feats, radius_per_query=rq, return_weights=True)
vq = rays["d"][:nq].repeat_interleave(S, dim=0).contiguous()
packed = dec._packed()
fn = lambda: point_ops.render_mlp(packed, pq, vq, npc.cloud_pos(), npc.col_feats, cg_, I_, w_, has_)
out = fn()
print("checksum", float(out.double().sum()))
for rep in range(3):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"three decoder kernels on 614,400 samples: {a.elapsed_time(b) / 10:.4f} ms")
bench.render_pass(npc, dec, ren, rays, dev)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    bench.render_pass(npc, dec, ren, rays, dev)
torch.cuda.synchronize()
print(f"frame {1e3 * (time.perf_counter() - t) / 5:.3f} ms")
