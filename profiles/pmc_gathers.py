"""Workload for the HBM-traffic (PMC) passes: GRU gate convolution, the two gather kernels, and a
calibration copy.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/fetch -o g -- python profiles/pmc_gathers.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/write -o g -- python profiles/pmc_gathers.py
(separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass, MI355X_MICROARCH.md)
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
g, video, graph = bench.build_graph(dev)
coords1, _ = video.reproject(graph.ii, graph.jj)
for _ in range(3):
    graph.corr(coords1, channels_last=True)
conv_launch, _ = bench.gru_gate_conv_workload(dev, graph.ii.shape[0], graph.ht, graph.wd,
                                              graph.ii if graph.share_context else None)
for _ in range(3):
    conv_launch()
npc, dec, ren, rays = bench.build_renderer(dev)
S = ren.N_surface
nq = 61440                                   # 96 image rows: the batch render_img evaluates
z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
rq = rays["radius"][:nq].repeat_interleave(S)
from glorie_slam_amd import point_ops  # noqa: E402
for _ in range(3):
    # the search as the renderer launches it (weights + mask from the same launch, bounded by the query radius)
    D, I, nn, w_, has_ = npc.index.search(pq, 8, radius_per_query=rq, image_layout=(S, rays["W"]), weights=(2, False, True))
    point_ops.idw_gather2(D, I, nn, npc.geo_feats, npc.col_feats, radius_per_query=rq)
# one full-frame render pass: the decoder kernels (the product's R2 gathers happen inside mlp_geo_v4 / mlp_nb_v4)
for _ in range(2):
    bench.render_pass(npc, dec, ren, rays, dev, two_streams=False)      # per-kernel figures: nothing beside them
# calibration: a 1 GiB float4 streaming copy (reads 1 GiB, writes 1 GiB)
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
for _ in range(3):
    dst = src.clone()
torch.cuda.synchronize()
