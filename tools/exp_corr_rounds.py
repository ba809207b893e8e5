"""Time of the fused displacement-major lookup + encoder against the number of edges (20 workgroups per edge at 60x80, 512
workgroup slots on the chip): shows the quantisation of the launch into workgroup rounds.  python tools/exp_corr_rounds.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench_corr import timeit  # noqa: E402
from glorie_slam_amd import droid_backends as db, update_ops as U  # noqa: E402

dev = torch.device("cuda", 0)
g, video, graph = bench.build_graph(dev)
coords1, _ = video.reproject(graph.ii, graph.jj)
N, h, w = int(graph.ii.shape[0]), graph.ht, graph.wd
arena = graph.corr
wgt = torch.randn(128, 196, 1, 1, device=dev) / 14
bias = torch.randn(128, device=dev)
w_dm = U.pack_corr_encoder_dm(wgt)
c = coords1.reshape(N, h, w, 2).float().contiguous()
for n in (8, 12, 16, 20, 24, 25, 26, 28, 32, 36):
    out = torch.empty(n, 128, h, w, dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last)
    cn, sl = c[:n].contiguous(), arena.slots[:n].contiguous()
    t = timeit(lambda: db.corr_dm_lookup(arena.views(), cn, h, w, slots=sl, interleaved=True, want_corr=False, enc_w=w_dm,
                                         enc_b=bias, enc_out=out))
    print(f"edges {n:3d}  workgroups {n * 20:4d}  {t:7.1f} us   {t / n:6.3f} us per edge")
