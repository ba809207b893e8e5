"""BA-update steps on graphs of 8 / 14 / 26 / 50 keyframes (the per-rank pose counts of the weak-scaling
bench at 1 / 2 / 4 / 8 GPUs), for  rocprofv3 --kernel-trace --stats  : how the solve grows with P."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    dev = "cuda:0"
    torch.cuda.set_device(0)
    for K in [int(a) for a in sys.argv[1:]] or [8, 14, 26, 50]:
        g, video, graph = bench.build_graph(dev, K=K)
        for i in range(3):
            graph.update(t0=1, t1=K, itrs=2, use_inactive=False)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for i in range(5):
            graph.update(t0=1, t1=K, itrs=2, use_inactive=False)
        t1.record(); torch.cuda.synchronize()
        print(f"K={K} edges={graph.ii.shape[0]} ms/step={t0.elapsed_time(t1) / 5:.3f}", flush=True)
        del g, video, graph
        torch.cuda.empty_cache()
