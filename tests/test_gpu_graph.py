"""Host-side mirrors on the GPU: FactorGraph (topology, update, update_lowmem) and DepthVideo."""
import numpy as np
import pytest
import torch

import glorie_slam_amd.synth as synth
from oracle import geom as ogeom, topology as otopo

pytestmark = pytest.mark.gpu


def _cfg(dev, H, W, buffer, ba="DSPO"):
    return {"cam": {"H_out": H, "W_out": W},
            "tracking": {"buffer": buffer, "backend": {"BA_type": ba}, "mono_thres": 0.1,
                         "multiview_filter": {"thresh": 0.01, "visible_num": 2}, "store_images": False},
            "device": str(dev), "setting": "t", "scene": "s", "data": {"output": "/tmp"}}


def make_video(dev, K, h, w, buffer=None, graph="keyframe", ba="DSPO"):
    from glorie_slam_amd.depth_video import DepthVideo
    g = synth.keyframe_graph(K=K, h=h, w=w, radius=3) if graph == "keyframe" else synth.loop_graph(K=K, h=h, w=w)
    fmaps, nets, inps = synth.feature_maps(K, h, w)
    video = DepthVideo(_cfg(dev, 8 * h, 8 * w, buffer or K, ba))
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    video.poses[:K] = t(g["poses"][:K])
    video.disps[:K] = t(g["disps"][:K])
    video.intrinsics[:] = t(g["intrinsics"][0])
    video.fmaps[:K] = t(fmaps)
    video.nets[:K] = t(nets)
    video.inps[:K] = t(inps)
    video.counter.value = K
    rng = np.random.default_rng(0)
    mono = g["disps"][:K] * rng.uniform(0.7, 1.4, (K, 1, 1)) + rng.uniform(-0.03, 0.03, (K, 1, 1))
    video.mono_disps[:K] = t(mono.astype(np.float32))
    return g, video


def make_graph(dev, video, corr_impl="volume", max_factors=-1):
    from glorie_slam_amd.factor_graph import FactorGraph
    from glorie_slam_amd.droid_net import UpdateModule
    torch.manual_seed(43)
    net = UpdateModule().to(dev).eval()
    return FactorGraph(video, net, device=str(dev), corr_impl=corr_impl, max_factors=max_factors)


def edges_of(graph):
    return list(zip(graph.ii.cpu().tolist(), graph.jj.cpu().tolist()))


def test_distance_matches_oracle_and_is_symmetric(gpu):
    g, video = make_video(gpu, 6, 24, 32)
    d = video.distance(beta=0.3).cpu().numpy()
    K = 6
    ii, jj = np.meshgrid(np.arange(K), np.arange(K), indexing="ij")
    d1 = ogeom.frame_distance(g["poses"], g["disps"], g["intrinsics"][0], ii.reshape(-1), jj.reshape(-1), 0.3)
    d2 = ogeom.frame_distance(g["poses"], g["disps"], g["intrinsics"][0], jj.reshape(-1), ii.reshape(-1), 0.3)
    np.testing.assert_allclose(d.reshape(-1), 0.5 * (d1 + d2), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(d, d.T, rtol=1e-6)


def test_topology_matches_literal_restatement(gpu):
    """bit-exact graph topology: the numpy topology code vs the loop-by-loop oracle, fed with the
    distances the HIP kernel produced"""
    g, video = make_video(gpu, 24, 16, 20, graph="loop")
    graph = make_graph(gpu, video, max_factors=60)
    graph.add_neighborhood_factors(0, 8, r=3)
    assert edges_of(graph) == otopo.neighborhood(0, 8, 3)
    # frontend proximity edges
    t = video.counter.value
    t0, t1 = t - 10, max(t - 20, 0)
    ii, jj = np.meshgrid(np.arange(t0, t), np.arange(t1, t), indexing="ij")
    d = video.distance(ii.reshape(-1), jj.reshape(-1), beta=0.75).cpu().numpy()
    before = edges_of(graph)
    want = otopo.proximity(d, t, before, t0=t0, t1=t1, rad=1, nms=1, thresh=16.0, max_factors=60)
    graph.add_proximity_factors(t0, t1, rad=1, nms=1, beta=0.75, thresh=16.0, remove=False)
    have = set(before)
    want_new = []
    for e in want:                      # add_factors drops duplicates, keeps order
        if e not in have:
            want_new.append(e)
    assert edges_of(graph)[len(before):] == want_new
    assert len(want_new) > 10
    # backend proximity edges on a fresh graph
    graph2 = make_graph(gpu, video, corr_impl="alt", max_factors=6 * t)
    ii, jj = np.meshgrid(np.arange(0, t), np.arange(0, t), indexing="ij")
    d = video.distance(ii.reshape(-1), jj.reshape(-1), beta=0.75).cpu().numpy()
    want = otopo.backend_proximity(d, 0, t, nms=5, radius=1, thresh=25.0, max_factors=6 * t)
    n = graph2.add_backend_proximity_factors(0, t, nms=5, radius=1, thresh=25.0, max_factors=6 * t, beta=0.75)
    seen, want_u = set(), []
    for e in want:
        if e not in seen:
            seen.add(e)
            want_u.append(e)
    assert edges_of(graph2) == want_u and n == len(want_u) and n > 2 * t


@pytest.mark.parametrize("corr_impl", ["volume", "otf"])
def test_update_runs_and_keeps_state_consistent(gpu, corr_impl):
    g, video = make_video(gpu, 6, 24, 32)
    graph = make_graph(gpu, video, corr_impl=corr_impl)
    graph.add_neighborhood_factors(0, 6, r=2)
    N = graph.ii.shape[0]
    p0 = video.poses.clone()
    for it in range(4):
        graph.update(t0=1, t1=6, itrs=2, opt_type="pose_depth" if it % 2 == 0 else "depth_scale")
    torch.cuda.synchronize()
    assert graph.net.shape == (1, N, 128, 24, 32) and graph.target.shape == (1, N, 24, 32, 2)
    assert torch.isfinite(video.poses).all() and torch.isfinite(video.disps).all()
    assert torch.equal(video.poses[0], p0[0])                       # pose 0 stays fixed
    assert (video.disps[:6] >= 1e-5).all()
    assert torch.isfinite(video.disps_up[:6]).all() and video.disps_up[:6].abs().sum() > 0
    assert (graph.age == 4).all()
    # quaternions stay unit length through the retractions
    assert torch.allclose(video.poses[:6, 3:].norm(dim=-1), torch.ones(6, device=gpu), atol=1e-4)


def test_volume_and_otf_updates_agree(gpu):
    """one update with the volume lookup vs the volume-free lookup: same flow targets within the
    fp16 tolerance of SURVEY 8(d)"""
    outs = []
    for impl in ("volume", "otf"):
        g, video = make_video(gpu, 5, 24, 32, ba="DBA")
        graph = make_graph(gpu, video, corr_impl=impl)
        graph.add_neighborhood_factors(0, 5, r=2)
        graph.update(t0=1, t1=5, itrs=2)
        outs.append((graph.target.clone(), video.poses.clone()))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=2e-2, atol=2e-2)
    assert torch.allclose(outs[0][1], outs[1][1], atol=1e-3)


def test_update_lowmem_backend_path(gpu):
    g, video = make_video(gpu, 10, 16, 24, graph="loop")
    graph = make_graph(gpu, video, corr_impl="alt", max_factors=60)
    n = graph.add_backend_proximity_factors(0, 10, nms=2, radius=1, thresh=50.0, max_factors=60, beta=0.75)
    assert n > 0
    graph.update_lowmem(t0=1, t1=10, itrs=2, steps=2)
    torch.cuda.synchronize()
    assert torch.isfinite(video.poses).all() and torch.isfinite(video.disps).all()
    assert graph.net.dtype == torch.float16 and torch.isfinite(graph.target).all()


def test_update_lowmem_chunking_and_correlation_operator(gpu):
    """one chunk over the whole graph == the reference's 8-source-frame chunks (all edges of a source frame
    share a chunk, GraphAgg is per source frame); the MFMA on-the-fly correlation stays within fp16 rounding
    of the fp32 alt-corr formulation"""
    outs = {}
    for name, chunk, corr in (("one", 1 << 30, "otf"), ("eight", 8, "otf"), ("alt", 8, "alt")):
        g, video = make_video(gpu, 20, 16, 24, graph="loop")
        graph = make_graph(gpu, video, corr_impl="alt", max_factors=200)
        graph.lowmem_chunk, graph.lowmem_corr = chunk, corr
        n = graph.add_backend_proximity_factors(0, 20, nms=2, radius=1, thresh=50.0, max_factors=200, beta=0.75)
        assert n > 0 and int(graph.ii.max()) >= 16          # more than two chunks of 8 source frames
        graph.update_lowmem(t0=1, t1=20, itrs=2, steps=1)
        outs[name] = [t.float().clone() for t in (graph.target, graph.weight, graph.net, video.disps[:20])]
    for a, b in zip(outs["one"], outs["eight"]):
        torch.testing.assert_close(a, b, atol=1e-3, rtol=1e-3)
    tgt_o, tgt_a = outs["eight"][0], outs["alt"][0]
    assert float((tgt_o - tgt_a).abs().max()) < 0.05          # flow targets in pixels


def test_rm_factors_and_rm_keyframe(gpu):
    g, video = make_video(gpu, 6, 16, 24, buffer=8)
    graph = make_graph(gpu, video)
    graph.add_neighborhood_factors(0, 6, r=2)
    n0 = graph.ii.shape[0]
    mask = (graph.ii == 5) | (graph.jj == 5)
    graph.rm_factors(mask, store=True)
    assert graph.ii.shape[0] == n0 - int(mask.sum()) and graph.ii_inac.shape[0] == int(mask.sum())
    assert len(graph.corr) == graph.ii.shape[0]               # the arena lists exactly the live edges
    assert len(set(graph.corr._host_slots)) == len(graph.corr)
    graph.rm_keyframe(3)
    assert int(graph.ii.max()) <= 3 and not ((graph.ii == 3) & (graph.jj == 3)).any()
    p0 = video.poses.clone()
    graph.update(t0=1, t1=4, itrs=1)
    assert torch.isfinite(video.poses).all() and not torch.equal(video.poses, p0)
    # a window that contains a frame without outgoing edges: eta has one row less than the BA has frames.  The
    # reference fails in ba_cuda (eta.view); the device would only set a status bit, so the host mirror raises
    with pytest.raises(RuntimeError, match="eta has"):
        graph.update(t0=1, t1=5, itrs=1)


def test_normalize_and_valid_mask(gpu):
    g, video = make_video(gpu, 7, 16, 24)
    p = video.poses.clone()
    s = video.disps[:7].mean().item()
    video.normalize()
    assert abs(video.disps[:7].mean().item() - 1.0) < 1e-5
    assert torch.allclose(video.poses[:7, :3], p[:7, :3] * s, rtol=1e-5)
    assert video.dirty[:7].all()
    video.update_valid_depth_mask(up=False)
    assert video.valid_depth_mask_small[:7].any()


def test_reprojection_writes_the_motion_features(gpu):
    """DepthVideo.reproject(motion=...) = reproject + FactorGraph._motion (factor_graph.py:205-221) in one launch: the pixel
    grid coords0 is (x, y) itself; same coordinates, same fp16 motion map"""
    from glorie_slam_amd.update_ops import PaddedFlow
    g, video = make_video(gpu, 6, 24, 32)
    graph = make_graph(gpu, video)
    graph.add_factors(torch.as_tensor(g["ii"], device=gpu), torch.as_tensor(g["jj"], device=gpu))
    n = int(graph.ii.shape[0])
    graph.target = (graph.target + 3.0 * torch.randn_like(graph.target)).contiguous()
    coords, _ = video.reproject(graph.ii, graph.jj)
    two = graph._motion(coords, padded=True)
    ref = two.buf.clone()
    pf = PaddedFlow(n, 24, 32, gpu)
    fused, _ = video.reproject(graph.ii, graph.jj, motion=(graph.target, pf))
    assert torch.equal(fused, coords)
    assert torch.equal(pf.buf, ref)


@pytest.mark.parametrize("stage2", [True, False])
def test_graph_replay_matches_eager(gpu, stage2):
    """use_graphs=True (hipGraph replay of update()) walks the same states as eager launches - also when every
    depth_scale stage falls back to pose_depth (depth_video.py:290-294): the replay defers that host decision to a
    flag the device stores into pinned memory"""
    outs, fallbacks = [], []
    for use_graphs in (False, True):
        g, video = make_video(gpu, 6, 24, 32)
        if stage2:
            video.cfg["tracking"]["multiview_filter"]["thresh"] = 0.25     # let stage 2 run (random-init operator)
        from glorie_slam_amd.factor_graph import FactorGraph
        from glorie_slam_amd.droid_net import UpdateModule
        torch.manual_seed(43)
        net = UpdateModule().to(gpu).eval()
        graph = FactorGraph(video, net, device=str(gpu), use_graphs=use_graphs)
        graph.add_factors(torch.as_tensor(g["ii"], device=gpu), torch.as_tensor(g["jj"], device=gpu))
        for i in range(8):
            graph.update(t0=1, t1=6, itrs=2, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
        if use_graphs:
            captured = [v for v in graph._graphs.values()
                        if isinstance(v, tuple) and isinstance(v[0], torch.cuda.CUDAGraph)]
            assert len(captured) == 2                                              # both stages replayed
            assert captured[0][1] is captured[1][1]                                # ... on shared static state
        outs.append([t.float().clone() for t in (video.poses, video.disps, video.disps_up, video.depth_scale,
                                                  graph.net, graph.target, graph.weight, graph.damping)])
        fallbacks.append(video.stage2_fallbacks)
    assert fallbacks[0] == fallbacks[1] == (0 if stage2 else 4)
    names = ["poses", "disps", "disps_up", "depth_scale", "net", "target", "weight", "damping"]
    for name, a, b in zip(names, *outs):
        assert torch.isfinite(a).all(), name
        torch.testing.assert_close(a, b, atol=2e-3, rtol=2e-3, msg=lambda m, n=name: f"{n}: {m}")
    # a topology change drops the captured graphs
    graph.rm_factors(graph.ii == 0, store=False)
    assert len(graph._graphs) == 0
    graph.update(t0=1, t1=6, itrs=2)


def test_capture_policy_and_shared_graph_pool(gpu):
    """capture_after = n: a call runs eagerly n times, is recorded on sighting n + 1 and replayed from then on (the frontend
    passes 6: its edge sets rarely live that long); every graph of a FactorGraph is recorded into ONE memory pool, so dropping
    and re-recording graphs (edge-set changes) does not grow the allocator's reservation"""
    from glorie_slam_amd.factor_graph import FactorGraph
    from glorie_slam_amd.droid_net import UpdateModule
    g, video = make_video(gpu, 6, 24, 32)
    torch.manual_seed(43)
    net = UpdateModule().to(gpu).eval()
    graph = FactorGraph(video, net, device=str(gpu), use_graphs=True, capture_after=3)
    ii, jj = torch.as_tensor(g["ii"], device=gpu), torch.as_tensor(g["jj"], device=gpu)
    graph.add_factors(ii, jj)
    seen = []
    for i in range(7):
        graph.update(t0=1, t1=6, itrs=2)
        seen.append((graph.stats["eager"], graph.stats["captures"], graph.stats["replays"]))
    # (a capture runs the eager code under stream capture: it counts as one eager pass and one replay of what it recorded)
    assert [s[1] for s in seen] == [0, 0, 0, 1, 1, 1, 1], seen
    assert seen[2][0] == 3 and seen[-1][2] >= 3, seen
    # edge-set changes: the graphs are dropped and recorded again, the pool's blocks are reused
    torch.cuda.synchronize()
    reserved = []
    for rep in range(6):
        graph.rm_factors(torch.ones_like(graph.ii, dtype=torch.bool), store=False)
        graph.add_factors(ii, jj)
        for i in range(5):
            graph.update(t0=1, t1=6, itrs=2)
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
    assert graph.stats["captures"] == 7
    assert reserved[-1] <= reserved[1] + (64 << 20), [r >> 20 for r in reserved]
    assert torch.isfinite(video.poses[:6]).all() and video.ctx().ba_status()[0] == 0


def test_full_resolution_valid_mask_and_video_npz(gpu, tmp_path):
    """update_valid_depth_mask(up=True) through glorie_valid_depth_mask == the
    reference's op-by-op formulation (global-memory radix-select median), and the video.npz format"""
    from oracle import se3 as ose3
    g, video = make_video(gpu, 6, 12, 16)
    video.cfg["tracking"]["multiview_filter"]["thresh"] = 0.05
    n = video.counter.value
    video.disps_up[:n] = torch.nn.functional.interpolate(video.disps[:n, None], scale_factor=8, mode="bilinear",
                                                         align_corners=False)[:, 0]
    video.timestamp[:n] = torch.arange(n, device=gpu).float() * 0.1
    masks = []
    for fused in (False, True):
        video.valid_depth_mask.zero_()
        video.dirty[:n] = True
        video.dirty[2] = False                                   # an index list with a hole
        video.update_valid_depth_mask(up=True, fused=fused)
        assert not bool(video.dirty.any())
        masks.append(video.valid_depth_mask[:n].clone())
    assert float((masks[0] != masks[1]).float().mean()) < 2e-3
    assert 0.05 < float(masks[1].float().mean()) < 1.0 and not bool(masks[1][2].any())
    # BA resolution through the same entry point
    small = []
    for fused in (False, True):
        video.valid_depth_mask_small.zero_()
        video.update_valid_depth_mask(up=False, fused=fused)
        small.append(video.valid_depth_mask_small[:n].clone())
    assert float((small[0] != small[1]).float().mean()) < 2e-3
    # camera-to-world pose = inverse of the stored world-to-camera [t, q]
    c2w = video.get_pose(3, "cpu").numpy()
    w2c = ose3.matrix(g["poses"][3])
    np.testing.assert_allclose(c2w @ w2c, np.eye(4), atol=1e-5)
    path = str(tmp_path / "video.npz")
    video.save_video(path)
    z = np.load(path)
    assert sorted(z.files) == ["depths", "poses", "timestamps", "valid_depth_masks"]
    assert z["poses"].shape == (n, 4, 4) and z["depths"].shape == (n, 96, 128) and z["timestamps"].shape == (n,)
    assert z["valid_depth_masks"].dtype == np.bool_ and z["valid_depth_masks"].shape == (n, 96, 128)
    np.testing.assert_allclose(z["depths"][1], 1.0 / video.disps_up[1].cpu().numpy(), rtol=1e-6)


def _replay_graph(gpu, use_graphs, K=6, h=24, w=32):
    from glorie_slam_amd.factor_graph import FactorGraph
    from glorie_slam_amd.droid_net import UpdateModule
    g, video = make_video(gpu, K, h, w)
    video.cfg["tracking"]["multiview_filter"]["thresh"] = 0.25
    torch.manual_seed(43)
    net = UpdateModule().to(gpu).eval()
    graph = FactorGraph(video, net, device=str(gpu), use_graphs=use_graphs)
    graph.add_factors(torch.as_tensor(g["ii"], device=gpu), torch.as_tensor(g["jj"], device=gpu))
    return video, graph


def test_graph_replay_with_inactive_factors_matches_eager(gpu):
    """the frontend's call (use_inactive=True, t0 = t1 = None: frontend.py:50-53) is captured too: (t0, t1) are
    resolved from host mirrors of the edge lists and the [inactive | active] buffers are static per edge set"""
    outs = []
    for use_graphs in (False, True):
        video, graph = _replay_graph(gpu, use_graphs, K=7)
        graph.rm_factors((graph.ii == 0) | (graph.jj == 0), store=True)        # frame 0's edges retire
        assert graph.ii_inac.numel() > 0
        for i in range(8):
            graph.update(None, None, use_inactive=True, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
        if use_graphs:
            captured = [v for v in graph._graphs.values()
                        if isinstance(v, tuple) and isinstance(v[0], torch.cuda.CUDAGraph)]
            assert len(captured) == 2
        assert graph._ba_args[7] == 2 and graph._ba_args[8] == 7             # t0 = max(1, ii.min() + 1), t1 = K
        outs.append([t.float().clone() for t in (video.poses, video.disps, video.disps_up, graph.net, graph.target,
                                                  graph.weight)])
    for name, a, b in zip(["poses", "disps", "disps_up", "net", "target", "weight"], *outs):
        assert torch.isfinite(a).all(), name
        torch.testing.assert_close(a, b, atol=2e-3, rtol=2e-3, msg=lambda m, n=name: f"{n}: {m}")


def test_graph_replay_survives_a_moved_scratch_arena(gpu):
    """growing the context's scratch arena re-allocates it; launches recorded into a hipGraph before that hold
    pointers into the freed block.  The context counts its moves (glorie_ctx_generation) and FactorGraph
    drops captured updates whose generation is stale instead of replaying them."""
    video_e, eager = _replay_graph(gpu, False)
    video_g, graph = _replay_graph(gpu, True)
    for i in range(4):
        for gr in (eager, graph):
            gr.update(t0=1, t1=6, itrs=2, opt_type="pose_depth")
    key = [k for k, v in graph._graphs.items() if isinstance(v, tuple) and isinstance(v[0], torch.cuda.CUDAGraph)]
    assert len(key) == 1
    ctx = video_g.ctx()
    gen0 = ctx.generation()
    ctx.reserve(1 << 28)                                     # e.g. the backend's global BA on the same video
    assert ctx.generation() == gen0 + 1
    junk = torch.full((1 << 24,), float("nan"), device=gpu)  # make re-use of the freed block likely
    for i in range(4):
        for gr in (eager, graph):
            gr.update(t0=1, t1=6, itrs=2, opt_type="pose_depth")
    assert graph._graphs[key[0]][-1] == (gen0 + 1, graph._arena_generations()[1])    # captured again
    torch.cuda.synchronize()
    del junk
    for a, b in ((video_e.poses, video_g.poses), (video_e.disps, video_g.disps), (eager.target, graph.target)):
        assert torch.isfinite(b).all()
        torch.testing.assert_close(a, b, atol=2e-3, rtol=2e-3)


def test_deferred_flag_resynchronises_after_a_missed_await(gpu):
    """the stage-1 decision of a replayed depth_scale stage travels through a pinned word tagged with a launch count
    (DepthVideo.publish_any_on / await_any_on).  If a replay's flag is never awaited the host's count falls behind the
    device's: the next await must warn, follow the device and keep working - not spin and raise on every later step."""
    _, video = make_video(gpu, 4, 12, 16)
    one = torch.ones(1, dtype=torch.int32, device=gpu)
    zero = torch.zeros(1, dtype=torch.int32, device=gpu)
    video.publish_any_on(one)
    assert video.await_any_on() == 1
    video.publish_any_on(zero)                    # a replay whose flag nobody reads ...
    video.publish_any_on(one)                     # ... and the next one
    with pytest.warns(UserWarning, match="resynchronised"):
        assert video.await_any_on(timeout=0.005) == 1
    video.publish_any_on(zero)                    # in step again: no warning, the right decisions
    assert video.await_any_on() == 0
    video.publish_any_on(one)
    assert video.await_any_on() == 1


def _load_topology_fixture():
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "topology.npz"))
    meta = json.loads(str(z["meta"]))
    return {k: (z["d_" + k], v) for k, v in meta.items()}


@pytest.mark.parametrize("name", sorted(_load_topology_fixture()))
def test_topology_matches_reference_fixture(gpu, name):
    """bit-exact graph topology against the REFERENCE: the scripts of tests/golden/recording.py replayed on the real
    DepthVideo + FactorGraph (HIP reproject, correlation arena with slot recycling for `volume`), `video.distance`
    answered from the stored matrix; edge lists, ages, inactive / bad lists, weight rows and proposal return values
    after every operation equal what /root/reference/src/factor_graph.py held (tests/golden/make_pins.py).  The CPU
    twin (fake buffers, also target / net rows) is tests/test_pins.py."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import recording as R
    d, meta = _load_topology_fixture()[name]
    K = d.shape[0]
    g, video = make_video(gpu, K, 16, 16, buffer=K + 2)
    dm = torch.from_numpy(d).to(gpu)
    video.distance = lambda ii=None, jj=None, beta=0.3, bidirectional=True: dm[
        torch.as_tensor(ii).long().reshape(-1).to(gpu), torch.as_tensor(jj).long().reshape(-1).to(gpu)].clone()
    graph = make_graph(gpu, video, corr_impl=meta["corr_impl"], max_factors=meta["max_factors"])
    states = R.run_topology_script(graph, video, meta["script"])
    keys = ("ii", "jj", "age", "ii_inac", "jj_inac", "ii_bad", "jj_bad", "weight_mean", "weight_inac_mean", "net_rows",
            "counter", "ret")
    for k, (got, want, op) in enumerate(zip(states, meta["states"], meta["script"])):
        if "cleared" in want:
            assert got == want
            continue
        for key in keys:
            assert got[key] == want[key], (name, k, op, key)
    if meta["corr_impl"] == "volume":
        assert len(graph.corr) == len(states[-1]["ii"])            # the arena holds exactly the live edges
