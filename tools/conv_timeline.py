"""Per-K-tile phase timeline of conv_igemm_kernel's waves (build conv.hip with -DEXP_CONV_STAMPS; tools/conv_timeline.sh).
Stamps of K-tiles 10-13 of every 61st workgroup; differences within a wave only (XCD clocks are not comparable)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = torch.zeros(16 * 4 * 4 * 8, dtype=torch.int64, device="cuda")
os.environ["GLORIE_CONV_STAMPS"] = str(stamps.data_ptr())
import bench  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "zr"
if which == "zr":
    launch, _ = bench.gru_gate_conv_workload(dev, 36, 60, 80)
else:
    from glorie_slam_amd import update_ops as U
    n, h, w, cin, nout = 36, 60, 80, 320, 128
    wp = U.pack_conv_igemm(torch.randn(nout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
    x = torch.randn(n, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
    out = torch.empty((n, nout, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
    launch = lambda: U.conv_igemm(x, None, wp, 9, nout, out)  # noqa: E731
for _ in range(3):
    launch()
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(16, 4, 4, 8)
ok = (s[..., :7] > 0).all(axis=(1, 2, 3))
s = s[ok]
names = ["wait for the DMA of this tile", "barrier A", "fragment reads", "barrier B", "DMA of the next tile issued", "MFMAs issued"]
d = np.diff(s[..., :7], axis=-1).reshape(-1, 6)
step = (s[:, :, 1:, 0] - s[:, :, :-1, 0]).reshape(-1)
print(f"{which}: {ok.sum()} workgroups x 4 waves x 4 K-tiles; cycles per K-tile: median {int(np.median(step))}  p10 "
      f"{int(np.percentile(step, 10))}  p90 {int(np.percentile(step, 90))}")
for k, nm in enumerate(names):
    print(f"   {nm:32s} median {int(np.median(d[:, k])):6d}   p10 {int(np.percentile(d[:, k], 10)):6d}   p90 {int(np.percentile(d[:, k], 90)):6d}")
tail = (s[:, :, 1:, 0] - s[:, :, :-1, 6]).reshape(-1)
print(f"   {'last MFMA issued -> next tile':32s} median {int(np.median(tail)):6d}   p90 {int(np.percentile(tail, 90)):6d}")
