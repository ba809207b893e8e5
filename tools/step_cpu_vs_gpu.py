"""Is the BA-update step CPU-launch bound?  Times the host-side issue loop and the device completion
separately.  Usage (GPU box): python tools/step_cpu_vs_gpu.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g, video, graph = bench.build_graph(dev, use_graphs="--graphs" in sys.argv)
    K = g["K"]
    n = [0]

    def step():
        opt = "pose_depth" if n[0] % 2 == 0 else "depth_scale"
        n[0] += 1
        graph.update(t0=1, t1=K, itrs=2, use_inactive=False, opt_type=opt)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    steps = 30
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host issue {1e3 * t_issue / steps:.3f} ms/step   issue+drain {1e3 * t_all / steps:.3f} ms/step")


if __name__ == "__main__":
    main()
