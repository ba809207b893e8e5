"""How fast is the KNN query kernel when the queries arrive sorted by grid cell (all lanes of a wave in the
same cell)?  Upper bound for a global query sort in front of the search."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = "cuda:0"
npc, dec, ren, rays = bench.build_renderer(dev)
S, nq = 10, 65536
z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
rq = rays["radius"][:nq].repeat_interleave(S)
idx = npc.index
cs = idx.cell_size
lo = npc.cloud_pos().min(0).values
key = ((pq - lo) / cs).floor().clamp_(min=0).long()
key = (key[:, 2] * 4096 + key[:, 1]) * 4096 + key[:, 0]


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"ray order      : {timed(lambda: idx.search(pq, 8, radius_per_query=rq)):8.1f} us")
order = torch.argsort(key)
ps, rs = pq[order].contiguous(), rq[order].contiguous()
print(f"cell order     : {timed(lambda: idx.search(ps, 8, radius_per_query=rs)):8.1f} us")
print(f"argsort(int64) : {timed(lambda: torch.argsort(key)):8.1f} us;  gather of the queries: "
      f"{timed(lambda: pq[order].contiguous()):.1f} us")
k32 = key.to(torch.int32)
print(f"argsort(int32) : {timed(lambda: torch.argsort(k32)):8.1f} us")
