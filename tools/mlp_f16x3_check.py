"""colour decoder: fp16 matrix cores with the 3-term split (mlp_col_v4) against the fp32 MFMA kernel (GLORIE_MLP_F32=1) on
the reference fixture, and the time of a render pass under both"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_render import _cfg, GOLD
from glorie_slam_amd.decoder import POINT
from glorie_slam_amd.neural_point import NeuralPointCloud

gpu = torch.device("cuda:0")
f = np.load(os.path.join(GOLD, "decoders.npz"))
t = lambda k: torch.from_numpy(f[k]).to(gpu)
torch.manual_seed(43)
dec = POINT(_cfg(gpu), c_dim=32, hidden_size=128, use_view_direction=True).eval().to(gpu)
npc = NeuralPointCloud(_cfg(gpu))
npc.add_points(t("cloud"), t("geo"), t("col"))
outs = {}
for mode in ("1", "0"):
    os.environ["GLORIE_MLP_F32"] = mode
    with torch.no_grad():
        raw, ray_mask, point_mask, counter = dec(t("p")[None], npc, "color", npc.geo_feats, npc.col_feats, pts_num=10,
                                                 cloud_pos=npc.cloud_pos(), pts_views_d=t("views"),
                                                 dynamic_r_query=t("radius"))
    outs[mode] = raw.cpu().numpy()
pm = f["point_mask"]
a, b = outs["1"][pm, :3], outs["0"][pm, :3]
print("rgb fp32-MFMA vs fixture  max abs %.3e" % np.abs(a - f["rgb"][pm]).max())
print("rgb f16x3     vs fixture  max abs %.3e" % np.abs(b - f["rgb"][pm]).max())
print("rgb f16x3 vs fp32-MFMA    max abs %.3e   (values in [%.3f, %.3f])" % (np.abs(a - b).max(), a.min(), a.max()))
oa, ob = outs["1"][pm, 3], outs["0"][pm, 3]
print("occ f16x3 vs fp32-MFMA    max abs %.3e   vs fixture %.3e (values in [%.2f, %.2f])" % (
    np.abs(oa - ob).max(), np.abs(ob - f["occ"][pm]).max(), oa.min(), oa.max()))
