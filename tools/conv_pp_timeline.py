"""Phase timeline of conv_pp_kernel's waves (build conv.hip with -DEXP_CONV_STAMPS; tools/conv_pp_timeline.sh).
Stamps of K-tiles 10-13 of every 41st workgroup: 0 behind the barrier that opens the wave's MEM phase, 1 behind the barrier
that opens its MFMA phase, 2 behind its last MFMA (+ DMA piece) issued.  Differences within a wave only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = torch.zeros(16 * 8 * 4 * 4, dtype=torch.int64, device="cuda")
os.environ["GLORIE_CONV_STAMPS"] = str(stamps.data_ptr())
from glorie_slam_amd import update_ops as U  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = 36, 60, 80
torch.manual_seed(1)
cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)  # noqa: E731
net, wide, pre = cl(128), cl(320), cl(256)
wzr = U.pack_conv_igemm(torch.randn(256, 320, 3, 3, device=dev) / 54, pair=True)
terms = torch.randn(n, 256, device=dev)
z, rnet = cl(128), cl(128)
for _ in range(3):
    U.conv_igemm(net, wide[:, 128:320], wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms, net=net, out2=rnet, pre=pre, policy="pp")
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(16, 8, 4, 4)
ok = (s > 0).all(axis=(1, 2, 3))
s = s[ok]
for g, nm in ((0, "group 0 (DMA behind its fragment reads)"), (1, "group 1 (DMA between its MFMAs)")):
    sg = s[:, 4 * g:4 * g + 4]
    step = (sg[:, :, 1:, 0] - sg[:, :, :-1, 0]).reshape(-1)
    mem = (sg[..., 1] - sg[..., 0]).reshape(-1)
    mfma = (sg[..., 2] - sg[..., 1]).reshape(-1)
    tail = (sg[:, :, 1:, 0] - sg[:, :, :-1, 2]).reshape(-1)
    pr = lambda v: f"median {int(np.median(v)):6d}   p10 {int(np.percentile(v, 10)):6d}   p90 {int(np.percentile(v, 90)):6d}"  # noqa: E731
    print(f"{nm}: {ok.sum()} workgroups x 4 waves; cycles per K-tile: {pr(step)}")
    own = (sg[..., 3] - sg[..., 0]).reshape(-1)
    print(f"   MEM phase: reads (+ DMA) issued, waits, barrier     {pr(mem)}")
    print(f"      of which own work (pieces issued, reads back)    {pr(own)}")
    print(f"   MFMA phase: 64 MFMAs (+ DMA) issued                 {pr(mfma)}")
    print(f"   last MFMA issued -> barrier -> next MEM phase       {pr(tail)}")
