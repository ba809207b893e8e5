"""s_memtime checkpoints of workgroup 0 of conv8_kernel (GLORIE_CONV8_DBG |= 128), K-tiles 10-12, waves 0 and 4"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = torch.zeros(8 * 128, dtype=torch.int64, device="cuda")
os.environ["GLORIE_CONV8_DBG"] = str(128 | int(os.environ.get("GLORIE_CONV8_DBG", "0")))
os.environ["GLORIE_CONV8_STAMPS"] = str(stamps.data_ptr())
from glorie_slam_amd import update_ops as U  # noqa: E402

dev = torch.device("cuda:0")
n, h, w, cin, nout = int(os.environ.get("MAPS", "36")), 60, 80, 448, 256
torch.manual_seed(0)
wp = U.pack_conv_igemm(torch.randn(nout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5)
x = torch.randn(n, cin, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
out = torch.empty((n, nout, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
for _ in range(3):
    U.conv_igemm(x, None, wp, 9, nout, out)
torch.cuda.synchronize()
s = stamps.cpu().view(8, 4, 4, 8)[:, :3]
names = ["phase start", "reads issued", "DMA issued", "vmcnt ok", "barrier 1", "lgkmcnt ok", "MFMAs issued", "barrier 2"]
t0 = int(s[0, 0, 0, 0])
for wv in (0, 4, 1, 5):
    print(f"wave {wv}: cycles since the previous checkpoint (tile 11), then per-phase total")
    for ph in range(4):
        row = s[wv, 1, ph]
        prev = int(s[wv, 1, ph - 1, 7]) if ph else int(s[wv, 0, 3, 7])
        d = [int(row[0]) - prev if ph == 0 else 0] + [int(row[k]) - int(row[k - 1]) for k in range(1, 8)]
        if ph:
            d[1] = int(row[1]) - prev
        print(f"  p{ph + 1}: " + "  ".join(f"{nm} {v:5d}" for nm, v in zip(names[1:], d[1:])) + f"   | {int(row[7]) - prev:5d}")
print("tile 10->11->12 start of wave 0:", [int(s[0, k, 0, 0]) - t0 for k in range(3)], " wave 4:", [int(s[4, k, 0, 0]) - t0 for k in range(3)])
