"""One mapping iteration on the 5000-ray batch in a loop (for rocprofv3 --kernel-trace --stats):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -- python tools/prof_train.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from glorie_slam_amd.render_train import FeatureAdam

dev = torch.device("cuda", 0)
npc, dec, ren, rays = bench.build_renderer(dev)
NR = int(os.environ.get("TRAIN_RAYS", 5000))          # 5000: mapper.py's batch; 1000: the batch of pipeline.SequenceRunner
pick = torch.randperm(rays["o"].shape[0], generator=torch.Generator().manual_seed(5))[:NR].to(dev)
b5 = {k: v[pick].contiguous() for k, v in rays.items() if torch.is_tensor(v)}
gt = torch.rand(NR, 3, device=dev)
hip = os.environ.get("TRAIN_TORCH") is None
geo = npc.geo_feats.detach().clone().requires_grad_(True)
col = npc.col_feats.detach().clone().requires_grad_(True)
opt = FeatureAdam([{"params": list(dec.parameters())}, {"params": [geo]}, {"params": [col]}])
ren.use_train_path, dec.use_fused = hip, hip


def it():
    opt.zero_grad()
    d, _, c, _, _ = ren.render_batch_ray(npc, dec, b5["d"], b5["o"], dev, "color", gt_depth=b5["depth"], npc_geo_feats=geo,
                                         npc_col_feats=col, cloud_pos=npc.cloud_pos(), dynamic_r_query=b5["radius"])
    (torch.abs(b5["depth"] - d).sum() + 0.5 * torch.abs(gt - c).sum()).backward()
    opt.step()


for _ in range(3):
    it()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    it()
torch.cuda.synchronize()
print("ms per iteration", 1e3 * (time.perf_counter() - t) / 10)
# host time of an iteration (enqueue only) and where it goes
import cProfile, pstats
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    it()
t_host = 1e3 * (time.perf_counter() - t) / 10
torch.cuda.synchronize()
print("host ms per iteration (enqueue only)", t_host)
if os.environ.get("TRAIN_CPROFILE"):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        it()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
