"""Frontend / Backend drivers on a synthetic keyframe video: the control flow of
frontend.py:40-131 and backend.py:27-98 on top of the HIP BA-update path.  The reference cannot run here
(CUDA only), so these are behavioural checks: bookkeeping of the window, the keyframe-redundancy branch,
edge budgets, fixed anchor frames, and finite state after every stage."""
import types

import numpy as np
import pytest
import torch

import glorie_slam_amd.synth as synth

pytestmark = pytest.mark.gpu


def _cfg(dev, H, W, buffer, keyframe_thresh=4.0, enable_loop=True, window=25):
    return {
        "cam": {"H_out": H, "W_out": W}, "device": dev, "setting": "test", "scene": "drivers",
        "data": {"output": "/tmp"},
        "tracking": {
            "buffer": buffer, "beta": 0.75, "warmup": 8, "max_age": 50, "mono_thres": 0.1,
            "multiview_filter": {"thresh": 0.25, "visible_num": 2}, "store_images": False,
            "frontend": {"enable_loop": enable_loop, "keyframe_thresh": keyframe_thresh, "thresh": 16.0,
                         "window": window, "radius": 1, "nms": 1, "max_factors": 75},
            "backend": {"BA_type": "DSPO", "thresh": 25.0, "radius": 1, "nms": 5, "normalize": False,
                        "loop_window": 25, "loop_thresh": 25.0, "loop_radius": 1, "loop_nms": 12},
        },
    }


def _video(dev, K, h, w, cfg, fill=None):
    from glorie_slam_amd.depth_video import DepthVideo
    fill = K if fill is None else fill
    g = synth.keyframe_graph(K=K, h=h, w=w, radius=3)
    fmaps, nets, inps = synth.feature_maps(K, h, w)
    video = DepthVideo(cfg)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    video.poses[:K] = t(g["poses"][:K])
    video.disps[:K] = t(g["disps"][:K])
    video.intrinsics[:] = t(g["intrinsics"][0])
    video.fmaps[:K] = t(fmaps)
    video.nets[:K] = t(nets)
    video.inps[:K] = t(inps)
    video.mono_disps[:K] = t(g["disps"][:K] * 0.8 + 0.01)
    video.timestamp[:K] = torch.arange(K, device=dev, dtype=torch.float)
    video.counter.value = fill
    return video, g


def _net(dev):
    from glorie_slam_amd.droid_net import UpdateModule
    torch.manual_seed(43)
    return types.SimpleNamespace(update=UpdateModule().to(dev).eval())


def _finite(video, K):
    return bool(torch.isfinite(video.poses[:K]).all() and torch.isfinite(video.disps[:K]).all()
                and (video.disps[:K] > 0).all())


def test_frontend_bootstrap_and_keyframe_branches(gpu):
    from glorie_slam_amd.frontend import Frontend
    h, w, K = 24, 32, 11
    cfg = _cfg(gpu, 8 * h, 8 * w, 16, keyframe_thresh=1e9, enable_loop=False)
    video, g = _video(gpu, K, h, w, cfg, fill=7)
    fe = Frontend(_net(gpu), video, cfg)
    fe()                                          # 7 of 8 warm-up frames: nothing happens
    assert not fe.is_initialized and fe.graph.ii.numel() == 0
    video.counter.value = 8
    pose0 = video.poses[0].clone()
    fe()                                          # bootstrap: neighbourhood graph, 16 pose_depth iterations
    assert fe.is_initialized and fe.t1 == 8
    assert torch.equal(video.poses[0], pose0)     # frame 0 is the gauge (t0 = 1)
    assert fe.graph.ii.numel() > 0 and int(fe.graph.ii.min()) >= cfg["tracking"]["warmup"] - 4
    assert fe.graph.ii_inac.numel() > 0           # the early edges were parked as inactive factors
    # set_dirty marks both flags; update_valid_depth_mask consumed `dirty`, `npc_dirty` is the mapper's
    assert bool(video.npc_dirty[:8].all()) and not bool(video.dirty[:8].any()) and _finite(video, 9)
    assert bool(video.valid_depth_mask[:8].any())
    assert torch.equal(video.poses[8], video.poses[7])   # initial guess for the next frame
    fe()                                          # no new keyframe: no-op
    assert fe.t1 == 8
    # a redundant keyframe (threshold huge): removed again, counter and window step back
    video.counter.value = 9
    n_before = fe.graph.ii.numel()
    fe()
    assert fe.t1 == 8 and video.counter.value == 8
    assert not bool(((fe.graph.ii == 8) | (fe.graph.jj == 8)).any())
    assert n_before <= fe.graph.ii.numel() <= cfg["tracking"]["frontend"]["max_factors"]
    # a useful keyframe (threshold 0): kept, 8 + 4 alternating DSPO iterations
    fe.keyframe_thresh = 0.0
    video.counter.value = 9
    fe()
    assert fe.t1 == 9 and video.counter.value == 9
    assert bool(((fe.graph.ii == 8) | (fe.graph.jj == 8)).any())
    assert fe.graph.ii.numel() <= cfg["tracking"]["frontend"]["max_factors"] + 4
    assert _finite(video, 10) and bool(video.npc_dirty[int(fe.graph.ii.min()):9].all())


def test_frontend_loop_closure_branch(gpu):
    """window smaller than the number of keyframes -> every kept keyframe calls loop_ba, which anchors the
    first frame of the loop window"""
    from glorie_slam_amd.frontend import Frontend
    h, w, K = 16, 20, 12
    cfg = _cfg(gpu, 8 * h, 8 * w, 16, keyframe_thresh=0.0, enable_loop=True, window=6)
    cfg["tracking"]["backend"]["loop_window"] = 6
    video, g = _video(gpu, K, h, w, cfg, fill=8)
    fe = Frontend(_net(gpu), video, cfg)
    fe()
    video.counter.value = 9
    fe()
    assert fe.t1 == 9 and getattr(fe, "last_loop_t", None) == 9
    assert _finite(video, 10)


def test_backend_dense_and_loop_ba(gpu):
    from glorie_slam_amd.backend import Backend
    from glorie_slam_amd.factor_graph import FactorGraph
    h, w, K = 16, 20, 14
    cfg = _cfg(gpu, 8 * h, 8 * w, 16)
    video, g = _video(gpu, K, h, w, cfg)
    net = _net(gpu)
    be = Backend(net, video, cfg)
    pose0 = video.poses[0].clone()
    video.npc_dirty[:] = False
    n, n_edges = be.dense_ba(steps=2)
    assert n == K and 0 < n_edges <= 2 * (1 + 2) * K
    assert torch.equal(video.poses[0], pose0) and _finite(video, K)
    assert bool(video.npc_dirty[:K].all())
    # loop BA seeded with a local graph: frames before the loop window stay put
    local = FactorGraph(video, net.update, device=gpu, corr_impl='volume', max_factors=48)
    local.add_neighborhood_factors(K - 5, K, r=2)
    be.backend_loop_window = 6
    before = video.poses[:K].clone()
    nk, ne = be.loop_ba(0, K, steps=2, local_graph=local)
    assert nk == 6 and ne >= 0
    assert torch.equal(video.poses[:K - 6 + 1], before[:K - 6 + 1])   # t0 = t_start_loop + 1
    assert _finite(video, K)
