"""2-rank (gloo, one GPU) walk of the bench's sharded graph: BA status word after every update"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
K = 6 * world + 2
g, video, graph = bench.build_graph(dev, K=K, h=60, w=80, rank=rank, world=world, use_graphs=True)
prev = None
for i in range(int(os.environ.get("NSTEPS", "60"))):
    opt = "pose_depth" if i % 2 == 0 else "depth_scale"
    graph.update(t0=1, t1=K, itrs=2, use_inactive=False, opt_type=opt)
    st = video.ctx().ba_status()
    if st != prev:
        print(f"rank {rank} step {i} {opt}: status {st} fallbacks {video.stage2_fallbacks} finite {bool(torch.isfinite(video.poses).all())}", flush=True)
        prev = st
dist.barrier()
dist.destroy_process_group()
