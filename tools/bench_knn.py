"""Times the neighbour search in query order vs image-patch order, and the one- vs two-table gather (bench scene)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from glorie_slam_amd import point_ops  # noqa: E402


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    npc, dec, ren, rays = bench.build_renderer(dev, 0, 1)
    S = ren.N_surface
    mode = os.environ.get("BENCH_KNN_MODE", "")
    for nq in ((61440,) if mode else (61440, 65536)):
        z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=dev)[None]
        pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
        rq = rays["radius"][:nq].repeat_interleave(S)
        if mode:                                 # one variant only (for counter passes)
            lay = (S, 640) if mode == "image" else None
            print(mode, timed(lambda: npc.index.search(pq, 8, radius_per_query=rq, image_layout=lay), 5), "ms")
            return
        t0 = timed(lambda: npc.index.search(pq, 8, radius_per_query=rq))
        t1 = timed(lambda: npc.index.search(pq, 8, radius_per_query=rq, image_layout=(S, 640)))
        D, I, nn = npc.index.search(pq, 8, radius_per_query=rq)
        g1 = timed(lambda: point_ops.idw_gather(D, I, nn, npc.geo_feats, radius_per_query=rq))
        g2 = timed(lambda: point_ops.idw_gather2(D, I, nn, npc.geo_feats, npc.col_feats, radius_per_query=rq))
        tot = t1 + g2
        print(f"rays {nq}: search plain {t0:.3f} ms, image {t1:.3f} ms; gather one table {g1:.3f} ms, two tables {g2:.3f} ms; "
              f"search+gather {tot:.3f} ms = {2156.0 * pq.shape[0] / tot / 1e6:.0f} GB/s ({2156.0 * pq.shape[0] / tot / 8e9 * 1e3:.3f} of 8 TB/s)",
              flush=True)


if __name__ == "__main__":
    main()
