"""Time of one rank's share of the 640x480 frame for world = 1, 2, 4, 8 (all on this GPU, one after the
other): the projected ray-shard speedup of the renderer, without the other ranks."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = "cuda:0"
base = None
for world in (1, 2, 4, 8):
    npc, dec, ren, rays = bench.build_renderer(dev, rank=0, world=world)
    for _ in range(2):
        bench.render_pass(npc, dec, ren, rays, dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        n = bench.render_pass(npc, dec, ren, rays, dev)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    base = base or ms
    print(f"world {world}: rays/rank {n}  {ms:.3f} ms per frame shard  -> speedup {base / ms:.2f}x", flush=True)
