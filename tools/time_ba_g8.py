import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from glorie_slam_amd import droid_backends as db
dev = torch.device("cuda:0")
g, video, graph = bench.build_graph(dev, K=8, use_graphs=False)
K = 8
for i in range(3):
    graph.update(t0=1, t1=K, itrs=2, use_inactive=False, opt_type="pose_depth")
tg, wg, dm, bi, bj = graph._ba_args[:5]
tg, wg = tg.reshape(-1, graph.ht, graph.wd, 2).contiguous(), wg.reshape(-1, graph.ht, graph.wd, 2).contiguous()
intr0 = video.intrinsics[0].contiguous()
pz, dz_ = video.poses.clone(), video.disps.clone()
def ba_only():
    pz.copy_(video.poses); dz_.copy_(video.disps)
    db.ba(pz, dz_, intr0, None, tg, wg, dm.reshape(-1, graph.ht, graph.wd), bi, bj, 1, K, 2, 1e-4, 0.1, False, False,
          ctx=video.ctx(), want_updates=False, targets_hwc=True)
for _ in range(10): ba_only()
res = []
for rep in range(5):
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    for _ in range(200): ba_only()
    b1.record(); torch.cuda.synchronize()
    res.append(b0.elapsed_time(b1) / 200 * 1e3)
print("G8 BA (2 GN iterations) us per call:", [round(r, 1) for r in res])
