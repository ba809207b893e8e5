"""conv8 timing vs number of workgroups (448->256, 60x80 maps): separates per-CU limits from chip-wide contention"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glorie_slam_amd import update_ops as U  # noqa: E402
from tools.bench_conv import timed  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    h, w, cin, nout = 60, 80, 448, 256
    torch.manual_seed(0)
    wt = torch.randn(nout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wp = U.pack_conv_igemm(wt)
    for n in [int(a) for a in sys.argv[1:]] or [1, 3, 6, 13, 27, 36, 41]:
        x = torch.randn(n, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
        out = torch.empty((n, nout, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
        t = timed(lambda: U.conv_igemm(x, None, wp, 9, nout, out))
        wgs = (n * h * w + 255) // 256
        fl = 2.0 * n * h * w * cin * 9 * nout
        print(f"maps {n:3d}: {wgs:4d} workgroups ({wgs / 256:.2f} per CU)  {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s  "
              f"{t / -(-wgs // 256):6.1f} us per round", flush=True)


if __name__ == "__main__":
    main()
