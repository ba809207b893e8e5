"""Three full-frame render passes (307,200 rays, 524k-point cloud) of the bench workload, for
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o r -- python tools/prof_render.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    dev = "cuda:0"
    torch.cuda.set_device(0)
    npc, dec, ren, rays = bench.build_renderer(dev)
    for _ in range(4):
        bench.render_pass(npc, dec, ren, rays, dev, two_streams=False)      # per-kernel figures: nothing beside them
    torch.cuda.synchronize()
