"""the search of NeuralPointCloud.add_neural_points on a cloud that covers part of the view: exact 8-NN (rounds 1-5) against the
ball-bounded search whose count it consumes (round 6):  python tools/exp_insert_search.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
npc, dec, ren, rays = bench.build_renderer(dev)
# surface points of a strided view (4800 rays), one half of them pushed off the cloud as a newly seen surface would be
sel = torch.arange(0, rays["o"].shape[0], 64, device=dev)
p = rays["o"][sel] + rays["d"][sel] * rays["depth"][sel, None]
p[::2] += torch.tensor([3.0, 0.5, -2.0], device=dev)
rad = rays["radius"][sel]
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t0 = timed(lambda: npc.index.search(p, 8, radius=npc.radius_add, radius_per_query=rad))
t1 = timed(lambda: npc.index.search(p, 8, radius=npc.radius_add, radius_per_query=rad, weights=(1, False, True)))
n0 = npc.index.search(p, 8, radius=npc.radius_add, radius_per_query=rad)[2]
n1 = npc.index.search(p, 8, radius=npc.radius_add, radius_per_query=rad, weights=(1, False, True))[2]
print(f"{p.shape[0]} insertion queries: exact 8-NN {t0:.3f} ms, ball-bounded {t1:.3f} ms, counts equal: {bool(torch.equal(n0, n1))}")
