"""The z|r gate launch (320 -> 256, 3x3, 36 x 60 x 80) on the staged 256-channel tile and on the weights-in-registers tile
(GLORIE_CONV_WREG, read per call): plain convolution, + bias + sigmoid, fused gate, fused gate with the shared context term.
Cases and modes are interleaved over several rounds (the first timings of a process run ~15 % slow), medians are reported.
    python tools/exp_conv_wreg.py [modes, default "0 1"]"""
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
pmap = (torch.arange(n, device=dev) % 8).int()
dynx = wide[:, 128:320]
wp = U.pack_conv_igemm(torch.randn(256, 320, 3, 3, device=dev) / (320 * 9) ** 0.5)
terms = torch.randn(n, 384, device=dev)
bias = torch.randn(256, device=dev)
fl = 2.0 * n * h * w * 320 * 9 * 256
modes = sys.argv[1:] or ["0", "1"]
bufs = {m: (cl(128), cl(128), cl(256), cl(128), cl(128)) for m in modes}
cases = {
    "plain": lambda b: U.conv_igemm(net, dynx, wp, 9, 256, b[2]),
    "bias+sigmoid": lambda b: U.conv_igemm(net, dynx, wp, 9, 256, b[2], terms=bias, act=U.ACT_SIGMOID),
    "gate": lambda b: U.conv_igemm(net, dynx, wp, 9, 256, b[0], epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256], net=net, out2=b[1]),
    "gate+ctx": lambda b: U.conv_igemm(net, dynx, wp, 9, 256, b[3], epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256], net=net,
                                       out2=b[4], pre=pre_kf[:, 0:256], pre_map=pmap),
}
times = {(m, c): [] for m in modes for c in cases}
for rnd in range(6):
    for c, fn in cases.items():
        for m in modes:
            os.environ["GLORIE_CONV_WREG"] = m
            t = B.timed(lambda: fn(bufs[m]), iters=10)
            if rnd:
                times[(m, c)].append(t)
med = lambda v: sorted(v)[len(v) // 2]
for m in modes:
    print(f"WREG={m}: " + " | ".join(f"{c} {med(times[(m, c)]):6.1f} us ({fl / med(times[(m, c)]) / 1e6:5.0f} TF/s)" for c in cases), flush=True)
if len(modes) > 1:
    print("identical:", all(torch.equal(a, b) for a, b in zip(bufs[modes[0]], bufs[modes[1]])))
