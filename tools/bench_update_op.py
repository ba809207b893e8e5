"""Times the update operator (row A4) in its three forms on the G8 shape: 36 edges, 60x80.
Usage (GPU box): python tools/bench_update_op.py [--edges 36] [--iters 30]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glorie_slam_amd.droid_net import UpdateModule, HalfUpdate, FusedUpdate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=36)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--form", default="all")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mod = UpdateModule().to(dev).eval()
    n, h, w = a.edges, 60, 80
    net = torch.tanh(torch.randn(1, n, 128, h, w, device=dev))
    inp = torch.relu(torch.randn(1, n, 128, h, w, device=dev))
    corr = torch.randn(1, n, 196, h, w, device=dev).half()
    flow = torch.randn(1, n, 4, h, w, device=dev)
    ii = torch.arange(n, device=dev) // 5
    uniq, ix = torch.unique(ii, return_inverse=True)
    groups = (ix, int(uniq.shape[0]))

    def autocast():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return mod(net, inp, corr, flow, ii, None, groups)

    forms = {"autocast": autocast,
             "half_cl": lambda f=HalfUpdate(mod): f(net, inp, corr, flow, ii, None, groups),
             "fused": lambda f=FusedUpdate(mod): f(net, inp, corr, flow, ii, None, groups)}
    for name, fn in forms.items():
        if a.form not in ("all", name):
            continue
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        print(f"{name:10s} {1e3 * (time.perf_counter() - t0) / a.iters:8.3f} ms / update", flush=True)


if __name__ == "__main__":
    main()
