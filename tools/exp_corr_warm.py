"""In-step cost of the fused lookup, cold vs warm: G8, eager BA-update steps; after each step the lookup is launched twice
with HIP events around each launch - the first finds the pyramid evicted by the step's ~2 GB of traffic (what the step's own
lookup sees), the second finds the 89 MB it reads in the Infinity Cache, at the same (in-step) clocks.
    python tools/exp_corr_warm.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glorie_slam_amd import update_ops as U  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    g, video, graph = bench.build_graph(dev)
    N = graph.ii.shape[0]
    wgt0 = torch.randn(128, 196, 1, 1, device=dev) / 14
    bias0 = torch.randn(128, device=dev)
    w_dm = U.pack_corr_encoder_dm(wgt0)
    c1 = torch.empty(N, 128, graph.ht, graph.wd, dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last)
    cold, warm = [], []
    for it in range(steps):
        graph.update(t0=1, t1=8, itrs=2, opt_type="pose_depth" if it % 2 == 0 else "depth_scale")
        coords1, _ = video.reproject(graph.ii, graph.jj)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        graph.corr.lookup_encode(coords1, w_dm, bias0, c1)
        ev[1].record()
        graph.corr.lookup_encode(coords1, w_dm, bias0, c1)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            cold.append(ev[0].elapsed_time(ev[1]) * 1e3)
            warm.append(ev[1].elapsed_time(ev[2]) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    alg = 936.0 * N * graph.ht * graph.wd
    for name, v in (("first launch after a step (cold pyramid)", cold), ("second launch (warm)", warm)):
        t = med(v)
        print(f"{name:44s} median {t:6.1f} us  min {min(v):6.1f}  frac {alg / t / 1e6 / 8.0:.3f}")


if __name__ == "__main__":
    main()
