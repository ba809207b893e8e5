import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = "cuda:0"
npc, dec, ren, rays = bench.build_renderer(dev, rank=0, world=8)
for _ in range(3):
    bench.render_pass(npc, dec, ren, rays, dev)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    bench.render_pass(npc, dec, ren, rays, dev)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
