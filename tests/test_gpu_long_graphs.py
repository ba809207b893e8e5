"""BASELINE configs 4 / 5 at their shapes: the global BA of the backend on a 128-keyframe 30x40 graph chosen by
add_backend_proximity_factors (ScanNet shape, config 4) and a 48x64 loop-closure graph (TUM shape, config 5), one
Gauss-Newton call of the device solver against the oracle restatement of ba_cuda; plus MotionFilter and
PoseTrajectoryFiller (the two remaining drivers of SURVEY 8(f) N3)."""
import types

import numpy as np
import pytest
import torch

import glorie_slam_amd.synth as synth
from oracle import ba as oba, geom as ogeom, se3 as ose3

pytestmark = pytest.mark.gpu


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _ba_vs_oracle(gpu, g, ii, jj, t0, t1, lm, ep, itrs=2):
    from glorie_slam_amd import droid_backends as db, _lib
    rng = np.random.default_rng(3)
    N = len(ii)
    K, h, w = g["K"], g["h"], g["w"]
    coords, _ = ogeom.reproject(g["poses"], g["disps"], g["intrinsics"], ii, jj)
    target = (coords.transpose(0, 3, 1, 2) + rng.normal(0, 0.5, (N, 2, h, w))).astype(np.float32)
    weight = rng.uniform(0, 1, (N, 2, h, w)).astype(np.float32)
    poses = g["poses"].copy()
    for k in range(1, K):
        poses[k] = ose3.retract((rng.standard_normal(6) * 0.003).astype(np.float32), poses[k])
    disps = (g["disps"] * (1 + 0.02 * rng.standard_normal(g["disps"].shape))).astype(np.float32)
    kx = sorted(set(list(range(t0, t1)) + [int(v) for v in ii]))
    eta = g["eta"][kx]
    rp, rd, rdx, rdz, info = oba.ba(poses, disps, g["intrinsics"][0], target, weight, eta, ii, jj, t0, t1, itrs, lm, ep)
    assert info["failed"] == 0
    p, d = _t(poses, gpu), _t(disps, gpu)
    dx, dz = db.ba(p, d, _t(g["intrinsics"][0], gpu), None, _t(target, gpu), _t(weight, gpu), _t(eta, gpu), _t(ii, gpu),
                   _t(jj, gpu), t0, t1, itrs, lm, ep, False, False)
    torch.cuda.synchronize()
    st = _lib.default_context().ba_status()
    assert st[0] == 0 and st[1] == len(kx), st
    np.testing.assert_allclose(dx.cpu().numpy(), rdx, rtol=1e-2, atol=2e-5)
    np.testing.assert_allclose(p.cpu().numpy(), rp, atol=1e-4)          # poses 1e-4
    np.testing.assert_allclose(d.cpu().numpy(), rd, atol=2e-4)
    return N


def test_global_ba_128_keyframes_config4_shape(gpu):
    """128 keyframes at 30x40 (ScanNet 240x320 / 8), edges from add_backend_proximity_factors like Backend.dense_ba:
    6 P = 762 unknowns -> the multi-launch blocked Cholesky; lm / ep of update_lowmem (factor_graph.py:306)"""
    from test_gpu_graph import make_video, make_graph
    K = 128
    g, video = make_video(gpu, K, 30, 40, graph="loop")
    graph = make_graph(gpu, video, corr_impl="alt", max_factors=6 * K)
    n = graph.add_backend_proximity_factors(0, K, nms=5, radius=1, thresh=25.0, max_factors=6 * K, beta=0.75)
    assert n > 3 * K
    ii, jj = graph.ii.cpu().numpy(), graph.jj.cpu().numpy()
    N = _ba_vs_oracle(gpu, g, ii, jj, 1, K, 1e-5, 1e-2)
    assert N == n


def test_loop_closure_ba_48x64_config5_shape(gpu):
    """TUM shape (384x512 / 8 = 48x64): sliding-window edges + loop-closure edges half a turn apart; 6 P = 234
    unknowns with a dense (non-banded) system -> the one-workgroup blocked Cholesky"""
    g = synth.loop_graph(K=40, h=48, w=64)
    _ba_vs_oracle(gpu, g, g["ii"], g["jj"], 1, g["K"], 1e-4, 0.1)


def _stream_cfg(dev, H, W, buffer):
    return {"cam": {"H_out": H, "W_out": W}, "device": str(dev), "setting": "t", "scene": "s", "data": {"output": "/tmp"},
            "mono_prior": {"predict_online": False}, "mapping": {"every_frame": 5},
            "tracking": {"buffer": buffer, "backend": {"BA_type": "DSPO"}, "mono_thres": 0.1,
                         "multiview_filter": {"thresh": 0.01, "visible_num": 2}, "store_images": True}}


def test_motion_filter_appends_keyframes(gpu):
    """MotionFilter.track (motion_filter.py:47-96): the first frame always becomes keyframe 0 (identity pose, unit
    disparity, features / context / mono prior stored); later frames only when the mean predicted flow exceeds the threshold"""
    from glorie_slam_amd.depth_video import DepthVideo
    from glorie_slam_amd.droid_net import DroidNet
    from glorie_slam_amd.motion_filter import MotionFilter
    H, W = 96, 128
    cfg = _stream_cfg(gpu, H, W, 8)
    video = DepthVideo(cfg)
    torch.manual_seed(43)
    net = DroidNet().to(gpu).eval()
    mono = lambda tstamp, image: torch.full((H, W), 2.0)
    mf = MotionFilter(net, video, cfg, thresh=1e9, device=str(gpu), mono_depth_fn=mono)
    g = torch.Generator().manual_seed(1)
    intr = torch.tensor([100.0, 100.0, 63.5, 47.5])
    img = lambda: torch.rand(1, 3, H, W, generator=g)
    assert mf.track(0, img(), intr) is True and video.counter.value == 1
    assert torch.equal(video.poses[0].cpu(), torch.tensor([0, 0, 0, 0, 0, 0, 1.0]))
    assert float(video.disps[0].min()) == 1.0 and torch.allclose(video.intrinsics[0].cpu(), intr / 8)
    assert float(video.fmaps[0].float().abs().sum()) > 0 and float(video.nets[0].float().abs().max()) <= 1.0
    assert float(video.inps[0].float().min()) >= 0.0 and torch.allclose(video.mono_disps[0], torch.full((12, 16), 0.5, device=gpu))
    assert mf.track(1, img(), intr) is False and video.counter.value == 1 and mf.count == 1     # threshold never reached
    mf.thresh = 0.0
    assert mf.track(2, img(), intr) is True and video.counter.value == 2 and mf.count == 0
    assert float(video.timestamp[1]) == 2.0 and float(video.fmaps[1].float().abs().sum()) > 0
    # the decision value is the mean flow norm of one update iteration at zero flow: same number from the plain module
    from glorie_slam_amd.droid_net import CorrBlock
    from glorie_slam_amd.factor_graph import coords_grid
    with torch.autocast("cuda"), torch.no_grad():
        corr = CorrBlock(mf.fmap[None, [0]], mf.fmap[None, [0]])(coords_grid(12, 16, device=gpu)[None, None])
        _, d_ref, _ = net.update(mf.net[None], mf.inp[None], corr)
    _, d_fused, _ = mf._fused(mf.net[None], mf.inp[None], corr)
    assert abs(float(d_ref.float().norm(dim=-1).mean()) - float(d_fused.float().norm(dim=-1).mean())) < 2e-2


def test_trajectory_filler_interpolates_and_refines(gpu):
    """PoseTrajectoryFiller (trajectory_filler.py:34-114): constant-velocity interpolation in the Lie algebra between the
    neighbouring keyframes, 12 motion-only updates on the parked frames, keyframes and counter untouched"""
    from glorie_slam_amd.depth_video import DepthVideo
    from glorie_slam_amd.droid_net import DroidNet
    from glorie_slam_amd.trajectory_filler import PoseTrajectoryFiller
    from glorie_slam_amd import lie
    h, w, K = 12, 16, 5
    H, W = 8 * h, 8 * w
    cfg = _stream_cfg(gpu, H, W, 16)
    video = DepthVideo(cfg)
    gsyn = synth.keyframe_graph(K=K, h=h, w=w, radius=2)
    fmaps, nets, inps = synth.feature_maps(K, h, w)
    video.poses[:K] = _t(gsyn["poses"][:K], gpu)
    video.disps[:K] = _t(gsyn["disps"][:K], gpu)
    video.intrinsics[:] = _t(gsyn["intrinsics"][0], gpu)
    video.fmaps[:K], video.nets[:K], video.inps[:K] = _t(fmaps, gpu), _t(nets, gpu), _t(inps, gpu)
    video.timestamp[:K] = torch.arange(K, device=gpu).float() * 10.0
    video.counter.value = K
    torch.manual_seed(43)
    net = DroidNet().to(gpu).eval()
    filler = PoseTrajectoryFiller(net=net, video=video, printer=None, device=str(gpu), batch=4, iters=12)
    # interpolation vs numpy: G(t) = exp(log(P1 P0^-1) (t - t0) / (t1 - t0 + 1e-3)) P0
    times = [5.0, 10.0, 17.5, 39.0, 45.0]
    Gs, t0, t1 = filler.interpolate(times)
    assert t0.tolist() == [0, 1, 1, 3, 4] and t1.tolist() == [1, 2, 2, 4, 4]
    P = lie.SE3(video.poses[:K])
    for n, t in enumerate(times):
        a, b = int(t0[n]), int(t1[n])
        xi = (P[b:b + 1] * P[a:a + 1].inv()).log() / (float(video.timestamp[b] - video.timestamp[a]) + 1e-3)
        ref = (lie.SE3.exp(xi * (t - float(video.timestamp[a]))) * P[a:a + 1]).data[0]
        assert torch.allclose(Gs.data[n], ref, atol=1e-5)
    assert torch.allclose(Gs.data[1], video.poses[1], atol=1e-5)         # exactly on a keyframe
    mid = 0.5 * (video.poses[0, :3] + video.poses[1, :3])
    assert torch.allclose(Gs.data[0, :3], mid, atol=2e-3)                # half way between keyframes 0 and 1

    class Stream:
        def __init__(self, n):
            g = torch.Generator().manual_seed(2)
            self.items = [(3.0 + 6.0 * k, torch.rand(1, 3, H, W, generator=g), None, None) for k in range(n)]

        def get_intrinsic(self):
            return torch.tensor([8 * 16.0, 8 * 16.0, 8 * 7.5, 8 * 5.5])

        def __iter__(self):
            return iter(self.items)

    kf = video.poses[:K].clone()
    out = filler(Stream(6))                                              # batches of 4 + 2
    assert out.data.shape == (6, 7) and bool(torch.isfinite(out.data).all())
    assert video.counter.value == K and torch.equal(video.poses[:K], kf)
    assert torch.allclose(out.data[:, 3:].norm(dim=-1), torch.ones(6, device=gpu), atol=1e-4)
    assert out.inv().matrix().shape == (6, 4, 4)                          # what slam.py:176-180 does with the result


@pytest.mark.parametrize("K", [256, 512])
def test_backend_dense_ba_beyond_128_keyframes_is_a_fixed_point(gpu, K):
    """BASELINE config 4 past the 128-keyframe test above: Backend.dense_ba (backend.py:27-62: proximity edges over the whole
    buffer, volume-free correlation, update_lowmem, 6 P up to 3066 unknowns -> the multi-launch blocked Cholesky) on a
    K-keyframe 30x40 loop trajectory.  No trained weights exist here, so the flow head's last layer is zeroed: the BA's
    targets are then the reprojections of the current state and the generating trajectory must stay where it is through the
    whole call - edge selection, 16-edge correlation batches, context terms per keyframe, the solver's size class."""
    import types
    from glorie_slam_amd.backend import Backend
    from glorie_slam_amd.depth_video import DepthVideo
    from glorie_slam_amd.droid_net import UpdateModule
    h, w = 30, 40
    cfg = {"cam": {"H_out": 8 * h, "W_out": 8 * w}, "device": str(gpu), "setting": "t", "scene": "s", "data": {"output": "/tmp"},
           "tracking": {"buffer": K + 8, "beta": 0.75, "warmup": 8, "max_age": 50, "mono_thres": 0.1,
                        "multiview_filter": {"thresh": 0.25, "visible_num": 2}, "store_images": False,
                        "frontend": {"enable_loop": False, "keyframe_thresh": 0.0, "thresh": 16.0, "window": 25,
                                     "radius": 1, "nms": 1, "max_factors": 75},
                        "backend": {"BA_type": "DSPO", "thresh": 25.0, "radius": 1, "nms": 5, "normalize": False,
                                    "loop_window": 25, "loop_thresh": 25.0, "loop_radius": 1, "loop_nms": 12}}}
    g = synth.loop_graph(K=K, h=h, w=w)
    fmaps, nets, inps = synth.feature_maps(K, h, w)
    video = DepthVideo(cfg)
    video.poses[:K] = _t(g["poses"][:K], gpu)
    video.disps[:K] = _t(g["disps"][:K], gpu)
    video.intrinsics[:] = _t(g["intrinsics"][0], gpu)
    video.fmaps[:K] = _t(fmaps, gpu)
    video.nets[:K] = _t(nets, gpu)
    video.inps[:K] = _t(inps, gpu)
    video.mono_disps[:K] = _t(g["disps"][:K], gpu)              # a prior that agrees with the state: stage 2 has nothing to pull
    video.counter.value = K
    torch.manual_seed(43)
    upd = UpdateModule().to(gpu).eval()
    with torch.no_grad():
        upd.delta[2].weight.zero_()
        upd.delta[2].bias.zero_()
    be = Backend(types.SimpleNamespace(update=upd), video, cfg)
    n, ne = be.dense_ba(steps=2)
    torch.cuda.synchronize()
    assert n == K and ne > 3 * K
    assert video.ctx().ba_status()[0] == 0
    p, p0 = video.poses[:K], _t(g["poses"][:K], gpu)
    assert bool(torch.isfinite(p).all() and torch.isfinite(video.disps[:K]).all())
    assert float((p[:, :3] - p0[:, :3]).norm(dim=-1).max()) < 2e-3
    assert float((1.0 - (p[:, 3:] * p0[:, 3:]).sum(-1).abs()).max()) < 1e-5
    rel = ((video.disps[:K] - _t(g["disps"][:K], gpu)).abs() / _t(g["disps"][:K], gpu)).mean()
    assert float(rel) < 0.02, float(rel)
