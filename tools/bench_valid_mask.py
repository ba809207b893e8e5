"""Times DepthVideo.update_valid_depth_mask(up=True) (8 keyframes, 480x640): fused HIP path against the
reference's op-by-op torch formulation.  Usage (GPU box): python tools/bench_valid_mask.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g, video, graph = bench.build_graph(dev)
    n = video.counter.value
    video.disps_up[:n] = torch.nn.functional.interpolate(video.disps[:n, None], scale_factor=8, mode="bilinear",
                                                         align_corners=False)[:, 0]
    for fused in (False, True):
        for _ in range(3):
            video.dirty[:n] = True
            video.update_valid_depth_mask(up=True, fused=fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            video.dirty[:n] = True
            video.update_valid_depth_mask(up=True, fused=fused)
        torch.cuda.synchronize()
        print(f"{'fused' if fused else 'torch'}: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms per update "
              f"({n} frames, {video.ht}x{video.wd}), valid {float(video.valid_depth_mask[:n].float().mean()):.3f}")


if __name__ == "__main__":
    main()
