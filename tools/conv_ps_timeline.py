"""K-tile timeline of conv_ps_kernel (producer / consumer form; build conv.hip with -DEXP_CONV_STAMPS, GLORIE_CONV_PS=1)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = torch.zeros(16 * 8 * 4 * 8, dtype=torch.int64, device="cuda")
os.environ["GLORIE_CONV8_STAMPS"] = str(stamps.data_ptr())
os.environ["GLORIE_CONV_PS"] = "1"
from glorie_slam_amd import update_ops as U  # noqa: E402

dev = torch.device("cuda:0")
n, h, w, cin, nout = 36, 60, 80, 320, 128
wp = U.pack_conv_igemm(torch.randn(nout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
x = torch.randn(n, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
out = torch.empty((n, nout, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
for _ in range(3):
    U.conv_igemm(x, None, wp, 9, nout, out)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(16, 8, 4, 8)[..., :4]
ok = (s > 0).all(axis=(1, 2, 3))
s = s[ok]
step = (s[:, :, 1:, 0] - s[:, :, :-1, 0]).reshape(-1)
print(f"{ok.sum()} workgroups; cycles per K-tile: median {int(np.median(step))}  p10 {int(np.percentile(step, 10))}  p90 "
      f"{int(np.percentile(step, 90))}")
for name, sl, names in (("consumer", slice(0, 4), ["wait at the barrier", "reads + 16 MFMAs (first half)", "reads + 16 MFMAs (second half)"]),
                        ("producer", slice(4, 8), ["wait for the DMA", "wait at the barrier", "next tile's DMA issued"])):
    d = np.diff(s[:, sl], axis=-1).reshape(-1, 3)
    for k, nm in enumerate(names):
        print(f"   {name}: {nm:34s} median {int(np.median(d[:, k])):6d}   p10 {int(np.percentile(d[:, k], 10)):6d}   p90 "
              f"{int(np.percentile(d[:, k], 90)):6d}")
