"""The z|r gate launch (320 -> 256, 3x3, 36 x 60 x 80) and the q gate launch (320 -> 128) with their epilogues taken apart:
plain convolution, + bias + sigmoid, fused gate, fused gate with the shared context term - on unpaired and paired weight
packings (update_ops.pack_conv_igemm(pair=...)).  Cases and packings are interleaved over several rounds (the first timings
of a process run ~15 % slow), medians are reported.  Build conv.hip with -DEXP_EPI_NO_LOADS / -DEXP_EPI_NO_STORE /
-DEXP_EPI_NO_SIGMOID (GLORIE_EXTRA_HIPFLAGS) for the ablation of the unpaired epilogue.
    python tools/bench_gate_epilogue.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_conv as B  # noqa: E402
from glorie_slam_amd import update_ops as U  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = 36, 60, 80
torch.manual_seed(1)
cl = lambda c, m=n: torch.randn(m, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
net, wide, pre_kf = cl(128), cl(320), cl(384, 8)
z0 = cl(128).abs().clamp(max=1.0)
pmap = (torch.arange(n, device=dev) % 8).int()
dynx = wide[:, 128:320]
wzr = torch.randn(256, 320, 3, 3, device=dev) / (320 * 9) ** 0.5
wq = torch.randn(128, 320, 3, 3, device=dev) / (320 * 9) ** 0.5
terms = torch.randn(n, 384, device=dev)
bias = torch.randn(256, device=dev)
fl = 2.0 * n * h * w * 320 * 9
modes = [False, True]
W = {m: (U.pack_conv_igemm(wzr, pair=m), U.pack_conv_igemm(wq, pair=m)) for m in modes}
bufs = {m: (cl(128), cl(128), cl(256), cl(128), cl(128), cl(128)) for m in modes}
cases = {
    "plain": (256, lambda m, b: U.conv_igemm(net, dynx, W[m][0], 9, 256, b[2])),
    "bias+sigmoid": (256, lambda m, b: U.conv_igemm(net, dynx, W[m][0], 9, 256, b[2], terms=bias, act=U.ACT_SIGMOID)),
    "z|r gate": (256, lambda m, b: U.conv_igemm(net, dynx, W[m][0], 9, 256, b[0], epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256],
                                                net=net, out2=b[1])),
    "z|r gate+ctx": (256, lambda m, b: U.conv_igemm(net, dynx, W[m][0], 9, 256, b[3], epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256],
                                                    net=net, out2=b[4], pre=pre_kf[:, 0:256], pre_map=pmap)),
    "q gate+ctx": (128, lambda m, b: U.conv_igemm(net, dynx, W[m][1], 9, 128, b[5], epilogue=U.EPI_GRU_Q, terms=terms[:, 256:],
                                                  net=net, z=z0, pre=pre_kf[:, 256:384], pre_map=pmap)),
}
times = {(m, c): [] for m in modes for c in cases}
for rnd in range(6):
    for c, (nout, fn) in cases.items():
        for m in modes:
            t = B.timed(lambda: fn(m, bufs[m]), iters=10)
            if rnd:
                times[(m, c)].append(t)
med = lambda v: sorted(v)[len(v) // 2]
for m in modes:
    print(f"pair={int(m)}: " + " | ".join(f"{c} {med(times[(m, c)]):6.1f} us ({fl * cases[c][0] / med(times[(m, c)]) / 1e6:5.0f} TF/s)"
                                           for c in cases), flush=True)
print("identical:", all(torch.equal(a, b) for a, b in zip(bufs[False], bufs[True])))
