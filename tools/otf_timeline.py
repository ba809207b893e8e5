"""s_memtime checkpoints of one workgroup of the otf8 lookup (GLORIE_OTF_DBG=32 in a -DEXP_OTF_DBG build of corr_otf.hip): where a workgroup's 40k cycles go"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = torch.zeros(4 * 32, dtype=torch.int64, device="cuda")
os.environ["GLORIE_OTF_DBG"] = "32"
os.environ["GLORIE_OTF_STAMPS"] = str(stamps.data_ptr())
import bench
from glorie_slam_amd.droid_net import OtfCorrBlock
dev = torch.device("cuda", 0)
g, video, graph = bench.build_graph(dev)
coords1, _ = video.reproject(graph.ii, graph.jj)
fm = video.fmaps
blk = OtfCorrBlock(fm.view(1, fm.shape[0] * fm.shape[1], *fm.shape[2:]))
wp = OtfCorrBlock.pack_encoder(torch.randn(128, 196, 1, 1, device=dev) / 14)
bias = torch.randn(128, device=dev)
hx = torch.zeros(36, 128, 60, 80, dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last)
for mode in ("plain", "fused"):
    for _ in range(3):
        if mode == "plain":
            blk(coords1, graph.ii, graph.jj)
        else:
            blk.lookup_encode(coords1, graph.ii, graph.jj, wp, bias, hx)
    torch.cuda.synchronize()
    s = stamps.cpu().view(4, 32)
    names = ["start", "prologue loaded"] + [f"L{l} {x}" for l in range(4) for x in ("mfma begin", "mfma end", "after bar", "extract end", "after bar")] + ["out begin", "end"]
    print(mode)
    for k, nm in enumerate(names):
        print(f"  {nm:18s}", "  ".join(f"{int(s[w, k] - s[0, 0]):7d}" for w in range(4)))
