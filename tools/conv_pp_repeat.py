"""Bitwise run-to-run repeatability of the two ping-pong convolution kernels (their LDS stages are guarded by barrier
counts and counted vmcnt waits - a protocol error shows up as a RARE wrong tile that comes and goes with load):

    python tools/conv_pp_repeat.py [reps] [--busy]

conv_pp_kernel (z|r gate launch) and conv_ppw_kernel (q gate, heads) on G8 (36 x 60 x 80) and on a 75-edge graph, every
repetition compared bit for bit with the first one AND with the default (non-ping-pong) kernel's result; --busy keeps a
second stream saturated with unrelated memory + matrix work (uneven load on the CUs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glorie_slam_amd import update_ops as U  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
busy = "--busy" in sys.argv
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
ba, bb = torch.randn(4096, 4096, device=dev, dtype=torch.float16), torch.randn(1 << 26, device=dev)


def kick():
    if busy:
        with torch.cuda.stream(side):
            torch.mm(ba, ba)
            bb.mul_(1.0001)


for n in (36, 75):
    h, w = 60, 80
    torch.manual_seed(n)
    cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)  # noqa: E731
    net, wide, pre = cl(128), cl(320), cl(384)
    dynx = wide[:, 128:320]
    z0 = cl(128).abs().clamp(max=1.0)
    wzr = U.pack_conv_igemm(torch.randn(256, 320, 3, 3, device=dev) / 54, pair=True)
    wq = U.pack_conv_igemm(torch.randn(128, 320, 3, 3, device=dev) / 54, pair=True)
    wh = U.pack_conv_igemm(torch.randn(384, 128, 3, 3, device=dev) / 34)
    tapw = U.pack_head_taps([torch.randn(2, 128, 3, 3, device=dev) / 30 for _ in range(2)])
    terms = torch.randn(n, 384, device=dev)
    bias = torch.randn(384, device=dev)
    out = lambda c: torch.empty((n, c, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)  # noqa: E731

    def zr(policy):
        z, rnet = out(128), out(128)
        U.conv_igemm(net, dynx, wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, :256], net=net, out2=rnet, pre=pre[:, :256],
                     policy=policy)
        return torch.cat([z, rnet], 1)

    def q(policy):
        new = out(128)
        U.conv_igemm(net, dynx, wq, 9, 128, new, epilogue=U.EPI_GRU_Q, terms=terms[:, 256:], net=net, z=z0, pre=pre[:, 256:384],
                     policy=policy)
        return new

    def heads(env):
        os.environ["GLORIE_CONV_PPW"] = env
        rest = out(128)
        rows = U.conv_igemm_heads(net, wh, 9, 384, bias, tapw, 2, out=rest)
        os.environ.pop("GLORIE_CONV_PPW", None)
        return torch.cat([rows.reshape(-1), rest.float().reshape(-1)])

    for name, fn, ref_arg, pp_arg in (("z|r gate (conv_pp_kernel)", zr, "nohalo", "pp"), ("q gate (conv_ppw_kernel)", q, "nohalo", "ppw"),
                                      ("heads (conv_ppw_kernel)", heads, "0", "1")):
        ref = fn(ref_arg).clone()
        first = fn(pp_arg).clone()
        bad = 0 if torch.equal(first, ref) else 1
        for _ in range(reps):
            kick()
            bad += 0 if torch.equal(fn(pp_arg), first) else 1
        torch.cuda.synchronize()
        print(f"{n} edges, {name}: {reps} repetitions{' beside a busy stream' if busy else ''}, {bad} differing "
              f"(first == default kernel: {bool(torch.equal(first, ref))})", flush=True)
