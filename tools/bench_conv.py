"""Times glorie_conv_igemm against MIOpen (F.conv2d, fp16 channels-last) on the update operator's
wide convolutions at the G8 shape (36 edges, 60x80).  Usage (GPU box): python tools/bench_conv.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glorie_slam_amd import update_ops as U  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    n, h, w = 36, 60, 80
    P = n * h * w
    torch.manual_seed(0)
    cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
    for (cin, nout, k) in [(448, 256, 3), (448, 128, 3), (128, 384, 3), (128, 128, 3), (128, 64, 3), (128, 128, 1)]:
        x = cl(cin)
        wt = (torch.randn(nout, cin, k, k, device=dev) / (cin * k * k) ** 0.5)
        wcl = wt.half().contiguous(memory_format=torch.channels_last)
        wp = U.pack_conv_igemm(wt)
        out = torch.empty((n, nout, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
        t_ref = timed(lambda: F.conv2d(x, wcl, padding=k // 2))
        t_own = timed(lambda: U.conv_igemm(x, None, wp, k * k, nout, out))
        err = float((out.float() - F.conv2d(x, wcl, padding=k // 2).float()).abs().max())
        fl = 2.0 * P * cin * k * k * nout
        print(f"{cin:4d}->{nout:4d} {k}x{k}: miopen {t_ref:7.1f} us ({fl / t_ref / 1e6:6.0f} TF/s)   "
              f"igemm {t_own:7.1f} us ({fl / t_own / 1e6:6.0f} TF/s)   max|diff| {err:.4f}", flush=True)


if __name__ == "__main__":
    main()


def gate_case():
    """the z|r gate launch of the update operator as the bench runs it: [net | corr, flow] 128 + 192 channels, hoisted
    context term, sigmoid / r * net epilogue - against the plain 320 -> 256 convolution of the same size"""
    dev = torch.device("cuda:0")
    n, h, w = 36, 60, 80
    torch.manual_seed(1)
    cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
    net, wide, pre = cl(128), cl(320), cl(384)
    dynx = wide[:, 128:320]
    wp = U.pack_conv_igemm(torch.randn(256, 320, 3, 3, device=dev) / (320 * 9) ** 0.5)
    terms = torch.randn(n, 384, device=dev)
    z, rnet, out = cl(128), cl(128), cl(256)
    fl = 2.0 * n * h * w * 320 * 9 * 256
    t_gate = timed(lambda: U.conv_igemm(net, dynx, wp, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256], net=net,
                                        out2=rnet, pre=pre[:, 0:256]))
    t_plain = timed(lambda: U.conv_igemm(net, dynx, wp, 9, 256, out))
    t_nopre = timed(lambda: U.conv_igemm(net, dynx, wp, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256], net=net,
                                         out2=rnet))
    bias = torch.randn(256, device=dev)
    t_sig = timed(lambda: U.conv_igemm(net, dynx, wp, 9, 256, out, terms=bias, act=U.ACT_SIGMOID))
    print(f"   gate without the context term {t_nopre:7.1f} us; plain + bias + sigmoid {t_sig:7.1f} us")
    print(f"gate 320->256: fused gate {t_gate:7.1f} us ({fl / t_gate / 1e6:6.0f} TF/s)   plain {t_plain:7.1f} us "
          f"({fl / t_plain / 1e6:6.0f} TF/s)", flush=True)


def tile_modes():
    """the z|r and q gate launches and the 128 -> 384 head convolution under the three tile policies of glorie_conv_igemm"""
    dev = torch.device("cuda:0")
    h, w = 60, 80
    torch.manual_seed(2)
    cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
    net, wide, pre = cl(128), cl(320), cl(384)
    dynx = wide[:, 128:320]
    wzr = U.pack_conv_igemm(torch.randn(256, 320, 3, 3, device=dev) / 54)
    wq = U.pack_conv_igemm(torch.randn(128, 320, 3, 3, device=dev) / 54)
    wh = U.pack_conv_igemm(torch.randn(384, 128, 3, 3, device=dev) / 34)
    terms = torch.randn(n, 384, device=dev)
    z, rnet, new, h1 = cl(128), cl(128), cl(128), cl(384)
    cases = {
        "z|r gate 320->256": lambda pol: U.conv_igemm(net, dynx, wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, 0:256],
                                                      net=net, out2=rnet, pre=pre[:, 0:256], policy=pol),
        "q gate 320->128": lambda pol: U.conv_igemm(rnet, dynx, wq, 9, 128, new, epilogue=U.EPI_GRU_Q, terms=terms[:, 256:384],
                                                    net=net, z=z, pre=pre[:, 256:384], policy=pol),
        "heads 128->384": lambda pol: U.conv_igemm(net, None, wh, 9, 384, h1, policy=pol),
    }
    for name, fn in cases.items():
        res = []
        for mode in ("128", "wide", "64", "split", "nohalo", None) + (("pp",) if "z|r" in name else ()):
            res.append(f"{mode or 'auto'} {timed(lambda: fn(mode)):6.1f} us")
        print(f"{name:20s} " + "   ".join(res), flush=True)


def pp_ab(rounds=6):
    """interleaved A/B of the z|r gate launch and the plain 448 -> 256 layer: default tile against the ping-pong tile"""
    dev = torch.device("cuda:0")
    n, h, w = 36, 60, 80
    torch.manual_seed(3)
    cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
    net, wide, pre = cl(128), cl(320), cl(256)
    dynx = wide[:, 128:320]
    wzr = U.pack_conv_igemm(torch.randn(256, 320, 3, 3, device=dev) / 54, pair=True)
    w448 = U.pack_conv_igemm(torch.randn(256, 448, 3, 3, device=dev) / 63)
    x448 = cl(448)
    terms = torch.randn(n, 256, device=dev)
    z, rnet, out = cl(128), cl(128), cl(256)
    gate = lambda pol: U.conv_igemm(net, dynx, wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms, net=net, out2=rnet, pre=pre,
                                    policy=pol)
    plain = lambda pol: U.conv_igemm(x448, None, w448, 9, 256, out, policy=pol)
    for name, fn, fl in (("z|r gate 320->256 (paired)", gate, 2.0 * n * h * w * 320 * 9 * 256),
                         ("plain 448->256", plain, 2.0 * n * h * w * 448 * 9 * 256)):
        ts = {"nohalo": [], "pp": []}                  # "nohalo" = the 256-channel x 128-pixel tile of conv_igemm_kernel for these layers
        for _ in range(rounds):
            for pol in ("nohalo", "pp"):
                ts[pol].append(timed(lambda: fn(pol), iters=20))
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        print(f"{name:28s} 256x128 {med['nohalo']:6.1f} us ({fl / med['nohalo'] / 1e6:5.0f} TF/s)   ping-pong {med['pp']:6.1f} us "
              f"({fl / med['pp'] / 1e6:5.0f} TF/s)   all 256x128 {[round(x, 1) for x in ts['nohalo']]} pp {[round(x, 1) for x in ts['pp']]}",
              flush=True)


if __name__ == "__main__" and os.environ.get("BENCH_CONV_GATE", "1") == "1":
    gate_case()
    tile_modes()
    pp_ab()


def ppw_ab(rounds=6, n=36):
    """interleaved A/B of conv_ppw_kernel (128 x 512 ping-pong tile) against the shipped kernels on the 128-channel layers of
    the update operator at G8: q gate (320 -> 128, blend epilogue, context term), heads (128 -> 384 + tap GEMMs),
    128 -> 128 3x3 (corr_encoder[1] / flow layers)"""
    import statistics
    dev = torch.device("cuda:0")
    h, w = 60, 80
    torch.manual_seed(2)
    cl = lambda c: torch.randn(n, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
    net, wide, pre = cl(128), cl(320), cl(128)
    dynx = wide[:, 128:320]
    z0 = cl(128).abs().clamp(max=1.0)
    wq = U.pack_conv_igemm(torch.randn(128, 320, 3, 3, device=dev) / (320 * 9) ** 0.5, pair=True)
    w384 = torch.randn(384, 128, 3, 3, device=dev) / (128 * 9) ** 0.5
    wh = U.pack_conv_igemm(w384)
    w128 = U.pack_conv_igemm(torch.randn(128, 128, 3, 3, device=dev) / (128 * 9) ** 0.5, pair=True)
    tapw = U.pack_head_taps([torch.randn(2, 128, 3, 3, device=dev) / 30 for _ in range(2)])
    terms = torch.randn(n, 128, device=dev)
    bias = torch.randn(384, device=dev)
    out = torch.empty_like(net)
    rest = torch.empty_like(net)

    def q_gate():
        U.conv_igemm(net, dynx, wq, 9, 128, out, epilogue=U.EPI_GRU_Q, terms=terms, net=net, z=z0, pre=pre)

    def heads():
        U.conv_igemm_heads(net, wh, 9, 384, bias, tapw, 2, out=rest)

    def c128():
        U.conv_igemm(net, None, w128, 9, 128, out, terms=bias[:128].contiguous(), act=U.ACT_RELU)

    cases = (("q gate 320->128", q_gate, 2.0 * n * h * w * 9 * 320 * 128), ("heads 128->384", heads, 2.0 * n * h * w * 9 * 128 * 384),
             ("128->128 3x3", c128, 2.0 * n * h * w * 9 * 128 * 128))
    res = {(c[0], e): [] for c in cases for e in ("0", "1")}
    for _ in range(rounds):
        for name, fn, _fl in cases:
            for e in ("0", "1"):
                os.environ["GLORIE_CONV_PPW"] = e
                res[(name, e)].append(timed(fn, 10))
    os.environ.pop("GLORIE_CONV_PPW", None)
    for name, _fn, fl in cases:
        a, b = statistics.median(res[(name, "0")]), statistics.median(res[(name, "1")])
        print(f"N={n} {name:18s} shipped {a:7.1f} us ({fl / a / 1e6:5.0f} TF/s)   128x512 ping-pong {b:7.1f} us ({fl / b / 1e6:5.0f} TF/s)", flush=True)


