"""Launch time of the fused displacement-major lookup against the number of edges: is the launch a sum of rounds of resident
waves (one tile per wave, 8 waves per CU -> 2048 tiles = 25.6 edges at 60x80 per round)?   python tools/exp_corr_rounds.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glorie_slam_amd import droid_backends as db, update_ops as U  # noqa: E402

dev = torch.device("cuda", 0)
g, video, graph = bench.build_graph(dev)
coords1, _ = video.reproject(graph.ii, graph.jj)
N, h, w = int(graph.ii.shape[0]), graph.ht, graph.wd
arena = graph.corr
w_dm = U.pack_corr_encoder_dm(torch.randn(128, 196, 1, 1, device=dev) / 14)
bias = torch.randn(128, device=dev)
c = coords1.reshape(N, h, w, 2).float().contiguous()
for n in (6, 12, 19, 25, 26, 32, 36, 44, 51, 52, 64, 76):
    idx = torch.arange(n, device=dev) % N
    cn, sl = c[idx].contiguous(), arena.slots[idx].contiguous()
    out = torch.empty(n, h, w, 128, dtype=torch.float16, device=dev)
    run = lambda: db.corr_dm_lookup(arena.views(), cn, h, w, slots=sl, interleaved=True, want_corr=False, enc_w=w_dm,
                                    enc_b=bias, enc_out=out.permute(0, 3, 1, 2))
    for _ in range(5):
        run()
    # 20 launches replayed from a hipGraph: the Python wrapper costs more than the kernel
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(20):
                run()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    tiles = n * ((h + 7) // 8) * ((w + 7) // 8)
    print(f"{n:3d} edges  {tiles:5d} tiles = {tiles / 2048:4.2f} rounds of 2048 waves   {us:6.1f} us   {us / tiles * 1000:6.2f} ns/tile")
