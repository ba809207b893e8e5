"""The collectives of glorie_slam_amd.dist through RCCL (torch.distributed backend "nccl") on the one GPU of
the test box: a world of ONE rank still initialises the communicator, runs ncclAllReduce / ncclAllGather on
device buffers and orders them against the HIP kernels on the stream.  (The N > 1 logic is covered by the gloo
tests - tests/test_dist_cpu.py, tests/test_gpu_sharded_update.py; two RCCL ranks cannot share one device.)"""
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


def _worker(rank, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        from glorie_slam_amd import _lib as L, dist as gdist, droid_backends as db
        from test_gpu_ba import make_problem
        res = {}
        for K, tag in ((6, "small"), (20, "packed")):          # 6P = 30 (dense exchange) and 114 (packed triangle)
            g = make_problem(K, 12, 16, radius=3)
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            # reference: the single-call BA
            p0, d0 = t(g["poses"]), t(g["disps"])
            db.ba(p0, d0, t(g["intrinsics"][0]), None, t(g["target"]), t(g["weight"]), t(g["eta"]), t(g["ii"]),
                  t(g["jj"]), 1, K, 2, 1e-4, 0.1, False, False)
            # the sharded form with its all-reduce forced through RCCL
            p1, d1 = t(g["poses"]), t(g["disps"])
            ctx = L.Context()
            gdist.ba_sharded(ctx, p1, d1, t(g["intrinsics"][0]), t(g["target"]), t(g["weight"]), t(g["eta"]),
                             t(g["ii"]), t(g["jj"]), 1, K, 2, 1e-4, 0.1, force_collective=True)
            torch.cuda.synchronize()
            res[tag] = (float((p0 - p1).abs().max()), float((d0 - d1).abs().max()), ctx.ba_status()[0])
        # pack -> all-reduce -> unpack leaves the lower triangle and v intact
        n6 = 120
        hv = torch.randn(n6 * n6 + n6, dtype=torch.float64, device=dev)
        ref = hv.clone()
        gdist.allreduce_system(hv, n6=n6, force=True)
        res["pack_roundtrip"] = bool(torch.equal(hv, ref))
        # owned-row exchange
        buf = torch.arange(12, dtype=torch.float32, device=dev).view(6, 2)
        want = buf.clone()
        gdist.allgather_owned_rows(buf, np.zeros(6, np.int64), 0, 1, force=True)
        res["rows"] = bool(torch.equal(buf, want))
        res["backend"] = dist.get_backend()
        torch.save(res, out_path)
    finally:
        dist.destroy_process_group()


def test_collectives_run_through_rccl(gpu, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(port, out), nprocs=1, join=True)
    res = torch.load(out)
    assert res["backend"] == "nccl"
    for tag in ("small", "packed"):
        dp, dd, st = res[tag]
        assert st == 0 and dp < 2e-6 and dd < 2e-6, (tag, res[tag])
    assert res["pack_roundtrip"] and res["rows"]


def _worker_native(rank, port, out_path):
    """the context-owned communicator (glorie_comm_init / glorie_allreduce_normal_eq / glorie_allgather_rows): no
    torch.distributed process group at all - the C ABI talks to RCCL itself"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from glorie_slam_amd import _lib as L, dist as gdist, droid_backends as db
    from test_gpu_ba import make_problem
    lib = L.load()
    res = {}
    ctx = L.Context()
    res["world_before"] = gdist.ctx_comm_world(ctx)
    hv0 = torch.ones(8, dtype=torch.float64, device=dev)
    res["rc_without_comm"] = int(lib.glorie_allreduce_normal_eq(ctx.handle, L.ptr(hv0), 8, L.stream_ptr()))
    res["world"] = gdist.init_ctx_comm(ctx)
    res["world_after"] = gdist.ctx_comm_world(ctx)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    for K, tag in ((6, "small"), (20, "packed")):
        g = make_problem(K, 12, 16, radius=3)
        p0, d0 = t(g["poses"]), t(g["disps"])
        db.ba(p0, d0, t(g["intrinsics"][0]), None, t(g["target"]), t(g["weight"]), t(g["eta"]), t(g["ii"]), t(g["jj"]), 1, K, 2,
              1e-4, 0.1, False, False)
        p1, d1 = t(g["poses"]), t(g["disps"])
        args = (t(g["intrinsics"][0]), t(g["target"]), t(g["weight"]), t(g["eta"]), t(g["ii"]), t(g["jj"]))
        gdist.ba_sharded(ctx, p1, d1, *args, 1, K, 2, 1e-4, 0.1)
        torch.cuda.synchronize()
        res[tag] = (float((p0 - p1).abs().max()), float((d0 - d1).abs().max()), ctx.ba_status()[0])
        # the same two iterations recorded into a hipGraph (kernels + the RCCL all-reduce as stream work) and replayed
        p2, d2 = t(g["poses"]), t(g["disps"])
        ps, ds = p2.clone(), d2.clone()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            gdist.ba_sharded(ctx, ps, ds, *args, 1, K, 2, 1e-4, 0.1)       # warm-up on the capture stream (arena sized)
        side.synchronize()
        ps.copy_(p2); ds.copy_(d2)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            gdist.ba_sharded(ctx, ps, ds, *args, 1, K, 2, 1e-4, 0.1)
        graph.replay()
        torch.cuda.synchronize()
        res[tag + "_graph"] = (float((ps - p1).abs().max()), float((ds - d1).abs().max()))
    # all-gather of owned rows through the C ABI (one rank: recv == send)
    send = torch.arange(64, dtype=torch.float32, device=dev)
    recv = torch.zeros_like(send)
    L.check(lib.glorie_allgather_rows(ctx.handle, L.ptr(send), L.ptr(recv), send.numel() * 4, L.stream_ptr()), "allgather")
    torch.cuda.synchronize()
    res["rows"] = bool(torch.equal(send, recv))
    L.check(lib.glorie_comm_destroy(ctx.handle), "glorie_comm_destroy")
    res["world_destroyed"] = gdist.ctx_comm_world(ctx)
    torch.save(res, out_path)


def test_context_owned_communicator_and_graph_capture(gpu, tmp_path):
    """glorie_allreduce_normal_eq on a ctx-owned RCCL communicator.  One rank here (two RCCL ranks cannot share
    a device): the communicator is created without torch.distributed, the sharded BA with the native exchange equals the
    single-call BA, and - what the torch.distributed form cannot do - build_system -> all-reduce -> solve_update replays
    from a hipGraph with the same result."""
    out = str(tmp_path / "native.pt")
    mp.spawn(_worker_native, args=(0, out), nprocs=1, join=True)
    res = torch.load(out)
    assert res["world_before"] == 0 and res["rc_without_comm"] != 0          # no communicator: an error, not a silent skip
    assert res["world"] == 1 and res["world_after"] == 1 and res["world_destroyed"] == 0
    for tag in ("small", "packed"):
        dp, dd, st = res[tag]
        assert st == 0 and dp < 2e-6 and dd < 2e-6, (tag, res[tag])
        gp, gd = res[tag + "_graph"]
        assert gp == 0.0 and gd == 0.0, (tag, res[tag + "_graph"])
    assert res["rows"]


def _worker_whole_step(rank, out_path, thresh):
    """a FactorGraph whose video is sharded over a (forced) world of ONE with the context's own communicator: every exchange
    of the step - normal equations, fallback flag, owned rows - goes through RCCL on the stream, and with use_graphs the
    whole step is one hipGraph"""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import bench
    from glorie_slam_amd import dist as gdist
    K = 9
    res = {}
    for tag, sharded, graphs in (("ref", False, False), ("eager", True, False), ("graph", True, True)):
        g, video, graph = bench.build_graph(dev, K=K, h=24, w=32, use_graphs=graphs)
        if thresh is not None:
            video.cfg["tracking"]["multiview_filter"]["thresh"] = thresh
        if sharded:
            owner = gdist.shard_frames(g["ii"], 1)
            video.enable_sharding(owner, 0, 1, force=True)
            res[tag + "_native"] = bool(video.native_exchange()) and gdist.ctx_comm_world(video.ctx()) == 1
        for i in range(8):
            graph.update(t0=1, t1=K, itrs=2, use_inactive=False, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
        stale_seq = None
        if sharded:
            # the host-side bookkeeping of a step must survive the replay: a consumer's fresh_disps_up() clears the flag, the
            # next (replayed) step rewrites this rank's disps_up rows and has to set it again (the replay path
            # returned before mark_upsampled, so the mapper / valid-depth mask / save_video read other ranks' stale rows)
            video.fresh_disps_up()
            a = bool(video.shard["stale_up"])
            replays0 = graph.stats["replays"]
            graph.update(t0=1, t1=K, itrs=2, use_inactive=False, opt_type="pose_depth")
            b = bool(video.shard["stale_up"])
            stale_seq = (a, b, graph.stats["replays"] - replays0)
            # (undo nothing: the reference run gets the same ninth update below)
        else:
            graph.update(t0=1, t1=K, itrs=2, use_inactive=False, opt_type="pose_depth")
        video.fresh_disps_up()
        torch.cuda.synchronize()
        res[tag] = {"stale_seq": stale_seq, "poses": video.poses[:K].cpu(), "disps": video.disps[:K].cpu(), "disps_up": video.disps_up[:K].cpu(),
                    "scale": video.depth_scale[:K].cpu(), "status": video.ctx().ba_status()[0],
                    "fallbacks": int(video.stage2_fallbacks), "stats": dict(graph.stats),
                    "whole_keys": sum(1 for k, v in graph._graphs.items()
                                      if isinstance(k, tuple) and len(k) == 10 and k[-2] and k[-1] and isinstance(v, tuple))}
    torch.save(res, out_path)


@pytest.mark.parametrize("thresh", [None, 1e-4])          # 1e-4: every depth_scale stage takes the stage-1 fallback
def test_whole_sharded_step_replays_from_one_graph(gpu, tmp_path, thresh):
    """with the context-owned communicator (the default whenever the process group runs on RCCL) the sharded
    step has no torch.distributed collective left in it: build -> all-reduce -> solve, the all-reduced fallback flag and the
    exchange of the owned rows are recorded with the update operator into ONE hipGraph per (edge set, stage).  One rank
    here (two RCCL ranks cannot share a device): the values must equal the unsharded step's, eagerly and replayed."""
    out = str(tmp_path / "whole.pt")
    mp.spawn(_worker_whole_step, args=(out, thresh), nprocs=1, join=True)
    res = torch.load(out)
    assert res["eager_native"] and res["graph_native"]
    ref = res["ref"]
    for tag in ("eager", "graph"):
        got = res[tag]
        assert got["status"] == 0 and got["fallbacks"] == ref["fallbacks"], (tag, got["fallbacks"], ref["fallbacks"])
        for name in ("poses", "disps", "disps_up", "scale"):
            torch.testing.assert_close(got[name], ref[name], atol=1e-4, rtol=1e-4, equal_nan=True, msg=lambda m, n=name, t=tag: f"{t} {n}: {m}")
    assert ref["fallbacks"] == (4 if thresh is not None else 0)
    assert res["eager"]["stale_seq"] == (False, True, 0)
    assert res["graph"]["stale_seq"] == (False, True, 1), res["graph"]["stale_seq"]     # set again by a REPLAYED step
    st = res["graph"]["stats"]
    assert st["captures"] == 2 and st["replays"] >= 4 and res["graph"]["whole_keys"] == 2, (st, res["graph"]["whole_keys"])
    for name in ("poses", "disps"):                            # a replayed step = the eager sharded step, bit for bit
        assert torch.equal(res["graph"][name], res["eager"][name]), name
