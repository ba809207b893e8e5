"""How many of the 8 neighbour slots carry a non-zero IDW weight on the bench's render workload, and how often a slot is
empty for ALL 16 samples a wave of the per-neighbour decoder owns (a wave-uniform skip of that neighbour's MLP pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
npc, dec, ren, rays = bench.build_renderer(dev)
S = ren.N_surface
nq = 61440
z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
rq = rays["radius"][:nq].repeat_interleave(S)
D, I, nn, w, has = npc.index.search(pq, 8, radius_per_query=rq, image_layout=(S, 640), weights=(2, False))
on = (w > 0)
print("samples", on.shape[0], "mean live slots per sample", float(on.sum(1).float().mean()), "has", float(has.float().mean()))
print("histogram of live slots:", torch.bincount(on.sum(1), minlength=9).tolist())
for group in (16, 32, 64):
    g = on.view(-1, group, 8).any(1)                     # slot k needed by a wave of `group` samples
    print(f"wave of {group} samples: slots needed per wave {float(g.sum(1).float().mean()):.2f} of 8")
