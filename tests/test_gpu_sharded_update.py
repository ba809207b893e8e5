"""The sharded FactorGraph.update (edges split by source keyframe over 2 processes, one all-reduce of the
normal equations per Gauss-Newton iteration, packed exchange of the owned disparity rows) walks the same
states as the single-process update.  Both ranks share the one GPU of the test box and talk over gloo; the
product runs the same code over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_updates(graph, K, n_updates):
    for i in range(n_updates):
        graph.update(t0=1, t1=K, itrs=2, use_inactive=False, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")


def _worker(rank, world, port, K, use_graphs, out_path, thresh=None):
    sys.path.insert(0, ROOT)
    import bench
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g, video, graph = bench.build_graph(torch.device("cuda", 0), K=K, h=24, w=32, rank=rank, world=world,
                                            use_graphs=use_graphs)
        if thresh is not None:
            video.cfg["tracking"]["multiview_filter"]["thresh"] = thresh
        _run_updates(graph, K, 6)
        video.fresh_disps_up()                      # collective: the deferred exchange of the upsampled rows
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"poses": video.poses[:K].cpu(), "disps": video.disps[:K].cpu(),
                        "disps_up": video.disps_up[:K].cpu(), "scale": video.depth_scale[:K].cpu(),
                        "status": video.ctx().ba_status(), "fallbacks": video.stage2_fallbacks}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("K,use_graphs", [(9, False), (9, True), (18, False)])   # K = 18: 6P = 102, packed triangle
def test_two_rank_update_matches_single_process(gpu, tmp_path, K, use_graphs):
    sys.path.insert(0, ROOT)
    import bench
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, port, K, use_graphs, out), nprocs=2, join=True)
    got = torch.load(out)
    g, video, graph = bench.build_graph(gpu, K=K, h=24, w=32, use_graphs=False)
    _run_updates(graph, K, 6)
    torch.cuda.synchronize()
    ref = {"poses": video.poses[:K].cpu(), "disps": video.disps[:K].cpu(), "disps_up": video.disps_up[:K].cpu(),
           "scale": video.depth_scale[:K].cpu()}
    for name in ("poses", "disps", "disps_up", "scale"):
        assert torch.isfinite(got[name]).all(), name
        # same arithmetic up to the summation order of the all-reduced fp64 system and fp16 convolutions on
        # differently sized batches
        torch.testing.assert_close(got[name], ref[name], atol=2e-3, rtol=2e-3, msg=lambda m, n=name: f"{n}: {m}")


def test_two_rank_stage1_fallback(gpu, tmp_path):
    """every depth_scale stage falls back to pose_depth (depth_video.py:290-294) on both ranks: the fallback BA of a shard
    spans the whole window, not the source frames its depth_scale stage saw - it needs its own damping rows (the device
    flags an eta / slot mismatch and skips the solve otherwise)"""
    sys.path.insert(0, ROOT)
    import bench
    K = 9
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, port, K, False, out, 1e-4), nprocs=2, join=True)
    got = torch.load(out)
    assert got["fallbacks"] == 3 and got["status"][0] == 0, (got["fallbacks"], got["status"])
    g, video, graph = bench.build_graph(gpu, K=K, h=24, w=32, use_graphs=False)
    video.cfg["tracking"]["multiview_filter"]["thresh"] = 1e-4
    _run_updates(graph, K, 6)
    torch.cuda.synchronize()
    assert video.stage2_fallbacks == 3
    for name, ref in (("poses", video.poses[:K].cpu()), ("disps", video.disps[:K].cpu())):
        torch.testing.assert_close(got[name], ref, atol=2e-3, rtol=2e-3, msg=lambda m, n=name: f"{n}: {m}")
