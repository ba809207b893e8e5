"""main-loop cycles per workgroup of conv8_kernel (s_memtime around the K loop of 64 workgroups, GLORIE_CONV8_DBG |= 256):
a clock-independent figure for A/B runs on different boxes.  448->256, 36 maps of 60x80: 63 K-tiles per workgroup."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = torch.zeros(1024 + 128, dtype=torch.int64, device="cuda")
base = int(os.environ.get("GLORIE_CONV8_DBG", "0"))
os.environ["GLORIE_CONV8_DBG"] = str(256 | base)
os.environ["GLORIE_CONV8_STAMPS"] = str(stamps.data_ptr())
from glorie_slam_amd import update_ops as U  # noqa: E402
from tools.bench_conv import timed  # noqa: E402

dev = torch.device("cuda:0")
n, h, w, cin, nout = int(os.environ.get("MAPS", "36")), 60, 80, int(os.environ.get("CIN", "448")), 256
torch.manual_seed(0)
wp = U.pack_conv_igemm(torch.randn(nout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
x = torch.randn(n, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
out = torch.empty((n, nout, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
t = timed(lambda: U.conv_igemm(x, None, wp, 9, nout, out))
torch.cuda.synchronize()
s = stamps.cpu()[1024:].view(64, 2)
d = (s[:, 1] - s[:, 0]).float()
T = 9 * cin // 64
print(f"dbg={base}: {t:7.1f} us; K loop {d.mean():9.0f} cycles per workgroup (min {d.min():.0f}, max {d.max():.0f}) = "
      f"{d.mean() / T:6.0f} per K-tile (MFMA floor 2176)", flush=True)
