"""Phase timeline of the fused displacement-major lookup's waves (build with -DEXP_DM_TIMESTAMPS: the kernel writes 8
shader-clock stamps over the first 64 bytes of each tile's first output row).  python tools/exp_corr_timeline.py [edges]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from glorie_slam_amd import droid_backends as db, update_ops as U  # noqa: E402

dev = torch.device("cuda", 0)
g, video, graph = bench.build_graph(dev)
coords1, _ = video.reproject(graph.ii, graph.jj)
N, h, w = int(graph.ii.shape[0]), graph.ht, graph.wd
arena = graph.corr
w_dm = U.pack_corr_encoder_dm(torch.randn(128, 196, 1, 1, device=dev) / 14)
bias = torch.randn(128, device=dev)
c = coords1.reshape(N, h, w, 2).float().contiguous()
names = ["unit+coords+setup, level-0 loads issued", "weights staged, barrier", "level 0 (+ loads of 1)", "level 1 (+ loads of 2)",
         "level 2 (+ loads of 3)", "level 3", "epilogue + stores landed"]
same_slot = os.environ.get("EXP_SAME_SLOT") == "1"          # every edge reads slot 0: the data stays in L2 / MALL
for n in [int(v) for v in sys.argv[1:]] or (8, 24, 36):
    out = torch.empty(n, h, w, 128, dtype=torch.float16, device=dev)
    cn, sl = c[:n].contiguous(), arena.slots[:n].contiguous()
    if same_slot:
        sl = torch.zeros_like(sl)
    cold = os.environ.get("EXP_COLD") == "1"                 # flush L2 / Infinity Cache in front of the stamped launch
    for rep in range(5):
        if cold and rep == 4:
            torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
        db.corr_dm_lookup(arena.views(), cn, h, w, slots=sl, interleaved=True, want_corr=False, enc_w=w_dm, enc_b=bias,
                          enc_out=out.permute(0, 3, 1, 2))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    st = np.stack([o[:, ty * 8, tx * 8, :52].copy().view(np.uint64).reshape(n, 13)
                   for ty in range((h + 7) // 8) for tx in range((w + 7) // 8) if ty * 8 < h and tx * 8 < w], 1)
    st = st.reshape(-1, 13).astype(np.int64)
    lat = st[:, 8] - st[:, 1]
    drain = st[:, 7] - st[:, 9]
    pre = np.stack([st[:, 10] - st[:, 0], st[:, 11] - st[:, 10], st[:, 12] - st[:, 11], st[:, 1] - st[:, 12]], 1)
    st = st[:, :8]
    d = np.diff(st, axis=1)
    life = st[:, 7] - st[:, 0]                  # (stamps of different XCDs are not comparable: only differences within a wave)
    print(f"== {n} edges, {st.shape[0]} waves; wave lifetime in shader clocks: median {int(np.median(life))}  p90 "
          f"{int(np.percentile(life, 90))}")
    print("   first phase: kernel arguments + unit + slot %d | coordinates fetched %d | level-0 state %d | first window's gathers issued %d"
          % tuple(int(v) for v in np.median(pre, 0)))
    print(f"   latency of the first window's gathers (issue done -> all landed): median {int(np.median(lat))}  p10 "
          f"{int(np.percentile(lat, 10))}  p90 {int(np.percentile(lat, 90))}")
    print(f"   of the last phase, waiting for the stores to land: median {int(np.median(drain))}  p90 {int(np.percentile(drain, 90))}")
    for k, nm in enumerate(names):
        print(f"   {nm:44s} median {int(np.median(d[:, k])):7d}   p90 {int(np.percentile(d[:, k], 90)):7d}")

# distinct 128-byte lines one gather instruction touches: lanes of a tile share a line when their displaced window origins agree
cc = c.cpu().numpy()
for l in range(4):
    hl, wl = h >> l, w >> l
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    bx = (np.floor(cc[..., 0] / 2 ** l).astype(np.int64) - 3 - (xs >> l)[None] + (wl >> 1)) % wl
    by = (np.floor(cc[..., 1] / 2 ** l).astype(np.int64) - 3 - (ys >> l)[None] + (hl >> 1)) % hl
    key = by * wl + bx
    cnt = []
    for ty in range(0, h, 8):
        for tx in range(0, w, 8):
            blk = key[:, ty:ty + 8, tx:tx + 8].reshape(N, -1)
            cnt.append([len(np.unique(r)) for r in blk])
    cnt = np.array(cnt)
    print(f"level {l}: distinct lines per gather instruction: mean {cnt.mean():.1f}  median {np.median(cnt):.0f}  p90 "
          f"{np.percentile(cnt, 90):.0f}  max {cnt.max()}")
