"""Workload of the correlation-lookup PMC passes (tools/pmc_corr.sh): on graph G8 (36 edges, 60x80)
  1. a 1 GiB device copy                                   (calibrates FETCH_SIZE / WRITE_SIZE on wide streaming access)
  2. the displacement-major lookup on a UNIFORM flow       (every line of a window is read exactly once by one wave
     instruction of 64 x 2 bytes: a known byte count in this kernel's own access pattern - the calibration
     MI355X_MICROARCH.md asks for before trusting an absolute FETCH_SIZE of another access width)
  3. the displacement-major lookup, lookup + fused encoder and the tiled lookup on the bench coordinates.
Launch order is fixed; tools/pmc_corr_summary.py reads the dispatches in that order."""
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glorie_slam_amd.droid_net import CorrArena  # noqa: E402
from glorie_slam_amd import update_ops as U  # noqa: E402

dev = torch.device("cuda:0")
g, video, graph = bench.build_graph(dev)
coords1, _ = video.reproject(graph.ii, graph.jj)
N, h, w = int(graph.ii.shape[0]), graph.ht, graph.wd
blk0, rig = graph._otf_block(), graph._otf_rig
arenas = {}
for lay in ("tiled", "dm"):
    arenas[lay] = CorrArena(h, w, dev, capacity=N, layout=lay)
    arenas[lay].add(blk0.levels[0], rig * graph.ii, rig * graph.jj)
wgt = torch.randn(128, 196, 1, 1, device=dev) / 14
bias = torch.randn(128, device=dev)
c1 = torch.empty(N, 128, h, w, dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last)
w_dm = U.pack_corr_encoder_dm(wgt)
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
torch.cuda.synchronize()
dst = src.clone()                                                   # 1: calibration copy
yy, xx = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32),
                        indexing="ij")
uniform = torch.stack([xx + 2.25, yy - 1.5], -1)[None, None].expand(1, N, h, w, 2).contiguous()
for _ in range(3):
    arenas["dm"].lookup_encode(uniform, w_dm, bias, c1)              # 2: uniform flow (fused form: 44 MB written)
for _ in range(3):
    arenas["dm"](coords1, channels_last=True)                       # 3a
for _ in range(3):
    arenas["dm"].lookup_encode(coords1, w_dm, bias, c1)             # 3b
for _ in range(3):
    arenas["tiled"](coords1, channels_last=True)                    # 3c
torch.cuda.synchronize()
# analytic line count of the uniform-flow launch: taps inside the level's plane, one 128-byte line per (tile, tap) with >= 1 live lane
import numpy as np
lines = 0
for l in range(4):
    hl, wl = h >> l, w >> l
    ys = np.floor((np.arange(h) - 1.5) / (1 << l)).astype(int) - 3
    xs = np.floor((np.arange(w) + 2.25) / (1 << l)).astype(int) - 3
    for ty in range((h + 7) // 8):
        for tx in range((w + 7) // 8):
            sy = np.arange(ty * 8, min(ty * 8 + 8, h)); sx = np.arange(tx * 8, min(tx * 8 + 8, w))
            s = set()
            for j in range(8):
                for i in range(8):
                    vy = (ys[sy] + j >= 0) & (ys[sy] + j < hl)
                    vx = (xs[sx] + i >= 0) & (xs[sx] + i < wl)
                    dy = (ys[sy] + j - (sy >> l))[vy]; dx = (xs[sx] + i - (sx >> l))[vx]
                    for a in set(dy.tolist()):
                        for b in set(dx.tolist()):
                            s.add((a, b))
            lines += len(s)
print("uniform_flow_lines", lines * N, "bytes", lines * N * 128)
