"""Correlation lookups on graph G8 (36 edges, 60x80) in isolation: the tiled volume gather, the volume-free MFMA
lookup and the lookup with the fused corr_encoder[0].
    python tools/bench_corr.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glorie_slam_amd.droid_net import OtfCorrBlock  # noqa: E402


def timeit(fn, reps=20, batches=5):
    """median over `batches` replays of a hipGraph holding `reps` back-to-back calls (us per call): device time,
    not the Python wrapper's launch rate"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda", 0)
    g, video, graph = bench.build_graph(dev)
    coords1, _ = video.reproject(graph.ii, graph.jj)
    N, HW = graph.ii.shape[0], graph.ht * graph.wd
    alg = 936.0 * N * HW
    from glorie_slam_amd.droid_net import CorrArena
    from glorie_slam_amd import update_ops as U
    blk0 = graph._otf_block()
    rig = graph._otf_rig
    c = (graph.ii == graph.jj).long()
    arenas = {}
    for lay in ("tiled", "dm"):
        arenas[lay] = CorrArena(graph.ht, graph.wd, dev, capacity=int(N), layout=lay)
        arenas[lay].add(blk0.levels[0], rig * graph.ii, rig * graph.jj + c)
    wgt0 = torch.randn(128, 196, 1, 1, device=dev) / 14
    bias0 = torch.randn(128, device=dev)
    c1 = torch.empty(N, 128, graph.ht, graph.wd, dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last)
    w_cl, w_dm = U.pack_corr_encoder(wgt0), U.pack_corr_encoder_dm(wgt0)
    t_vol = timeit(lambda: arenas["tiled"](coords1, channels_last=True))
    t_vol_enc = timeit(lambda: U.conv_igemm(arenas["tiled"](coords1, channels_last=True), None, w_cl, 1, 128, c1,
                                            terms=bias0, act=U.ACT_RELU))
    t_dm = timeit(lambda: arenas["dm"](coords1, channels_last=True))
    t_dm_enc = timeit(lambda: arenas["dm"].lookup_encode(coords1, w_dm, bias0, c1))
    for name, t in (("tiled gather (channels-last)", t_vol), ("tiled gather + 1x1 encoder launch", t_vol_enc),
                    ("displacement-major gather", t_dm), ("displacement-major gather + encoder", t_dm_enc)):
        print(f"{name:38s} {t:8.1f} us   {alg / t / 1e6:7.2f} TB/s of algorithmic bytes   frac {alg / t / 1e6 / 8.0:.3f}")
    if os.environ.get("CORR_ONLY_VOLUME"):
        return
    fm = video.fmaps
    blk = OtfCorrBlock(fm.view(1, fm.shape[0] * fm.shape[1], *fm.shape[2:]))
    t_otf = timeit(lambda: blk(coords1, graph.ii, graph.jj))
    wgt = torch.randn(128, 196, 1, 1, device=dev) / 14
    bias = torch.randn(128, device=dev)
    wp = OtfCorrBlock.pack_encoder(wgt)
    hx = torch.zeros(N, 320, graph.ht, graph.wd, dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last)
    t_enc = timeit(lambda: blk.lookup_encode(coords1, graph.ii, graph.jj, wp, bias, hx[:, 128:256]))
    t_enc2 = timeit(lambda: blk.lookup_encode(coords1, graph.ii, graph.jj, wp, bias, hx[:, 128:256]))
    t_otf2 = timeit(lambda: blk(coords1, graph.ii, graph.jj))
    print("second pass: fused", t_enc2, "plain", t_otf2)
    for name, t in (("tiled volume gather", t_vol), ("volume-free MFMA lookup", t_otf), ("lookup + fused corr_encoder[0]", t_enc)):
        print(f"{name:34s} {t:8.1f} us   {alg / t / 1e6:7.2f} TB/s of algorithmic bytes   frac {alg / t / 1e6 / 8.0:.3f}")


if __name__ == "__main__":
    main()
