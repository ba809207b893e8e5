"""K-tile timeline of conv_ps_kernel (producer / consumer form; build conv.hip with -DEXP_CONV_STAMPS, GLORIE_CONV_PS=1)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = torch.zeros(16 * 8 * 4 * 8, dtype=torch.int64, device="cuda")
os.environ["GLORIE_CONV8_STAMPS"] = str(stamps.data_ptr())
mode = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["GLORIE_CONV_PS"] = mode
from glorie_slam_amd import update_ops as U  # noqa: E402

dev = torch.device("cuda:0")
n, h, w, cin, nout = 36, 60, 80, 320, int(sys.argv[2]) if len(sys.argv) > 2 else 128
wp = U.pack_conv_igemm(torch.randn(nout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
x = torch.randn(n, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
out = torch.empty((n, nout, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
for _ in range(3):
    U.conv_igemm(x, None, wp, 9, nout, out)
torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(16, 8, 4, 8)
if mode == "1":
    groups = (("consumer", slice(0, 4), 4, ["wait at the barrier", "reads + 16 MFMAs (first half)", "reads + 16 MFMAs (second half)"]),
              ("producer", slice(4, 8), 4, ["wait for the DMA", "wait at the barrier", "next tile's DMA issued"]))
else:
    groups = (("consumer", slice(0, 4), 4, ["second half's reads + 32 MFMAs (first half)", "wait for all reads + barrier",
                                            "next first half's reads + 32 MFMAs"]),
              ("producer", slice(4, 8), 4, ["wait for the DMA (tile t)", "wait at the barrier", "DMA of tile t + 2 issued"]))
ok = (raw[:, :, :, :4] > 0).all(axis=(1, 2, 3))
print(f"GLORIE_CONV_PS={mode}, 320->{nout}: {ok.sum()} workgroups sampled")
for name, sl, k, names in groups:
    s = raw[ok][:, sl, :, :k]
    step = (s[:, :, 1:, 0] - s[:, :, :-1, 0]).reshape(-1)
    print(f"   {name}: cycles per K-tile: median {int(np.median(step))}  p10 {int(np.percentile(step, 10))}  p90 "
          f"{int(np.percentile(step, 90))}")
    d = np.diff(s, axis=-1).reshape(-1, k - 1)
    for j, nm in enumerate(names):
        print(f"      {nm:40s} median {int(np.median(d[:, j])):6d}   p10 {int(np.percentile(d[:, j], 10)):6d}   p90 "
              f"{int(np.percentile(d[:, j], 90)):6d}")
