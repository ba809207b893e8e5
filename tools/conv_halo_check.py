"""(record of a removed experiment) halo-staged 3x3 convolutions (GLORIE_CONV_HALO=2 in the build that had them) against the
per-tap staging: same bits, and the time of the step's layers.  With the current library both columns are the per-tap kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glorie_slam_amd import update_ops as U
from tools.bench_conv import timed

dev = torch.device("cuda:0")
torch.manual_seed(5)


def cases(n, h, w):
    cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
    net, wide, pre = cl(128), cl(320), cl(384)
    dynx = wide[:, 128:320]
    wzr = U.pack_conv_igemm(torch.randn(256, 320, 3, 3, device=dev) / 54)
    wq = U.pack_conv_igemm(torch.randn(128, 320, 3, 3, device=dev) / 54)
    wh = U.pack_conv_igemm(torch.randn(384, 128, 3, 3, device=dev) / 34)
    w2 = U.pack_conv_igemm(torch.randn(128, 128, 3, 3, device=dev) / 34)
    terms = torch.randn(n, 384, device=dev)
    b = torch.randn(384, device=dev)
    z, rnet, new, h1, o = cl(128), cl(128), cl(128), cl(384), cl(128)
    return {
        "z|r gate 320->256": (lambda: U.conv_igemm(net, dynx, wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256],
                                                   net=net, out2=rnet, pre=pre[:, 0:256]), (z, rnet)),
        "q gate 320->128": (lambda: U.conv_igemm(net, dynx, wq, 9, 128, new, epilogue=U.EPI_GRU_Q, terms=terms[:, 256:384],
                                                 net=net, z=z, pre=pre[:, 256:384]), (new,)),
        "heads 128->384": (lambda: U.conv_igemm(net, None, wh, 9, 384, h1, terms=b, act=U.ACT_RELU), (h1,)),
        "ce2 128->128": (lambda: U.conv_igemm(net, None, w2, 9, 128, o, terms=b[:128].contiguous(), act=U.ACT_RELU), (o,)),
    }


for (n, h, w) in [(3, 10, 13), (5, 12, 16), (36, 60, 80)]:
    cs = cases(n, h, w)
    for name, (fn, outs) in cs.items():
        res = {}
        for mode in ("0", "2"):
            os.environ["GLORIE_CONV_HALO"] = mode
            for o in outs:
                o.zero_()
            fn()
            torch.cuda.synchronize()
            res[mode] = [o.clone() for o in outs]
        same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(res["0"], res["2"]))
        line = f"{n}x{h}x{w} {name:20s} identical {same}"
        if n == 36:
            ts = []
            for mode in ("0", "2"):
                os.environ["GLORIE_CONV_HALO"] = mode
                ts.append(timed(fn))
            line += f"   per-tap {ts[0]:6.1f} us   halo {ts[1]:6.1f} us"
        print(line, flush=True)
