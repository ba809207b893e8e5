"""Kernel time of one rank's 1/8 share of the 640x480 frame against the wall time of the pass: is the small shard's
overhead launch gaps (what a hipGraph of the batch would remove) or the kernels themselves at 1/8 of the work?
    rocprofv3 --kernel-trace --stats -d gpurun_out/shard8 -- python tools/render_shard_kernels.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = "cuda:0"
world = int(os.environ.get("SHARD_WORLD", "8"))
npc, dec, ren, rays = bench.build_renderer(dev, rank=0, world=world)
for _ in range(3):
    bench.render_pass(npc, dec, ren, rays, dev)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    n = bench.render_pass(npc, dec, ren, rays, dev)
torch.cuda.synchronize()
print(f"world {world}: rays/rank {n}  {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per pass (23 passes in the trace)")
