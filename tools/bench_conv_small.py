"""tile policies of glorie_conv_igemm on the step's short-K / narrow convolutions (36 x 60 x 80): python tools/bench_conv_small.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glorie_slam_amd import update_ops as U
from tools.bench_conv import timed

dev = torch.device("cuda:0")
n, h, w = 36, 60, 80
torch.manual_seed(3)
cl = lambda c, m=n: torch.randn(m, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
x256, x128, o128, o64 = cl(256), cl(128), cl(128), cl(64)
b128, b64 = torch.randn(128, device=dev), torch.randn(64, device=dev)
w_ce1 = U.pack_conv_igemm(torch.randn(128, 256, 1, 1, device=dev) / 16)
w_ce2 = U.pack_conv_igemm(torch.randn(128, 128, 3, 3, device=dev) / 34)
w_fe2 = U.pack_conv_igemm(torch.randn(64, 128, 3, 3, device=dev) / 34)
cases = {
    "ce1 1x1 256->128": lambda pol: U.conv_igemm(x256, None, w_ce1, 1, 128, o128, terms=b128, act=U.ACT_RELU, policy=pol),
    "ce2 3x3 128->128": lambda pol: U.conv_igemm(x128, None, w_ce2, 9, 128, o128, terms=b128, act=U.ACT_RELU, policy=pol),
    "fe2 3x3 128->64": lambda pol: U.conv_igemm(x128, None, w_fe2, 9, 64, o64, terms=b64, act=U.ACT_RELU, policy=pol),
}
for name, fn in cases.items():
    res = []
    for mode in ("128", "wide", "64", None):
        res.append(f"{mode or 'auto'} {timed(lambda: fn(mode)):6.1f} us")
    print(f"{name:20s} " + "   ".join(res), flush=True)
