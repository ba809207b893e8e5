"""One whole `FactorGraph.update()` on the GPU against the composition of the oracle pieces
(oracle/update_step.py: reproject -> lookup -> update operator -> BA / DSPO stage 2 -> upsampling, with the
bookkeeping of /root/reference/src/factor_graph.py:212-256 between them), at 30x40 and at the BASELINE size
G8 (8 keyframes, 36 edges, 60x80).

Two comparisons per stage:

  * `stagewise`: the oracle BA / stage 2 / upsampling are fed with the flow targets, confidences, damping
    and upsampling mask the GPU produced in that call.  Tolerances are the tight ones of SURVEY 8(d): poses
    1e-4, disparities 2e-4 - this pins the dispatcher, the [N,h,w,2] target layout, `damping = 0.2 eta + EP`,
    the `t0` rule, the clamp, which frames are upsampled, `age`.
  * `end to end`: the oracle runs the update operator itself as the plain fp32 module (pinned to the
    reference by tests/golden/update_module.npz).  The GPU evaluates that operator in fp16 like the
    reference's autocast, so flow revisions carry fp16 noise (flows rel 2e-2 / abs 1e-2); a
    disparity reacts to the flow noise of ITS pixel (dz ~ d(flow) / Jz), so disparities get 5e-3 here.
"""
import numpy as np
import pytest
import torch

from oracle import update_step as ostep, geom as ogeom
from test_gpu_graph import make_video, make_graph

pytestmark = pytest.mark.gpu

MV_THRESH = 0.25      # two-view threshold that lets stage 2 run with a random-init operator (bench.py make_cfg)


def _np(t):
    return t.detach().float().cpu().numpy()


def _video_state(video):
    n = video.counter.value
    return dict(poses=_np(video.poses), disps=_np(video.disps), disps_up=_np(video.disps_up),
                intrinsics=_np(video.intrinsics), mono_disps=_np(video.mono_disps),
                depth_scale=_np(video.depth_scale), depth_shift=_np(video.depth_shift),
                valid_small=video.valid_depth_mask_small.cpu().numpy().copy(), n=n)


def _cfg(video):
    mv = video.cfg["tracking"]["multiview_filter"]
    return dict(mv_thresh=mv["thresh"], visible_num=mv["visible_num"], mono_thres=video.mono_thres)


def _setup(gpu, K, h, w, seed=5):
    g, video = make_video(gpu, K, h, w)
    video.cfg["tracking"]["multiview_filter"]["thresh"] = MV_THRESH
    graph = make_graph(gpu, video)
    graph.add_neighborhood_factors(0, K, r=3)
    rng = np.random.default_rng(seed)
    N = graph.ii.shape[0]
    graph.target = graph.target + torch.from_numpy(rng.normal(0, 0.5, (1, N, h, w, 2)).astype(np.float32)).to(gpu)
    graph.weight = torch.from_numpy(rng.uniform(0, 1, (1, N, h, w, 2)).astype(np.float32)).to(gpu)
    return g, video, graph


def _graph_state(video, graph):
    """the oracle's own copy of the edge state, built from the keyframe buffers (add_factors,
    factor_graph.py:95-143): net/inp of the source frame, fp16 pyramid of <fmap_i/4, fmap_j/4>"""
    ii, jj = graph.ii.cpu().numpy(), graph.jj.cpu().numpy()
    fm = video.fmaps[:, 0].cpu().numpy()
    return dict(ii=ii, jj=jj, net=_np(video.nets[graph.ii]), inp=_np(video.inps[graph.ii]),
                target=_np(graph.target[0]), weight=_np(graph.weight[0]), damping=_np(graph.damping),
                age=graph.age.cpu().numpy().copy(), pyramid=ostep.corr_pyramid_fp16(fm[ii], fm[jj]))


def _fp32_update_fn():
    """the update operator as a plain fp32 torch module on the host (same seed-43 default init as the
    module inside `make_graph`; its outputs are pinned to the reference by test_golden.py)"""
    from glorie_slam_amd.droid_net import UpdateModule
    torch.manual_seed(43)
    mod = UpdateModule().eval()

    def fn(net, inp, corr, motn, ii, jj):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()[None]
        with torch.no_grad():
            o_net, delta, weight, eta, up = mod(t(net), t(inp), t(corr), t(motn), torch.from_numpy(ii),
                                                torch.from_numpy(jj))
        # the reference's autocast hands the upsampling mask over in fp16 (droid_net.py:62-66)
        return o_net[0].numpy(), delta[0].numpy(), weight[0].numpy(), eta[0].numpy(), up[0].half().numpy()
    return fn


@pytest.mark.parametrize("K,h,w", [(6, 30, 40), (8, 60, 80)])
def test_update_stagewise_matches_oracle(gpu, K, h, w):
    g, video, graph = _setup(gpu, K, h, w)
    cfg = _cfg(video)
    for it, opt in enumerate(["pose_depth", "depth_scale", "pose_depth"]):
        before = _video_state(video)
        age0 = graph.age.clone()
        graph.update(t0=None if it == 0 else 1, t1=K, itrs=2, EP=1e-7, opt_type=opt)
        torch.cuda.synchronize()
        target, weight, damping, ii, jj, uniq, upmask, t0, t1 = graph._ba_args
        if not torch.is_tensor(upmask):
            upmask = upmask.evaluate()          # FusedUpdate hands DepthVideo.upsample the unevaluated logits
        # bookkeeping of factor_graph.py:229-256
        assert t0 == max(1, int(graph.ii.min()) + 1) == 1 and t1 == K
        assert torch.equal(graph.age, age0 + 1)
        assert torch.equal(uniq, torch.unique(graph.ii))
        eta_gpu = graph.damping[uniq]
        torch.testing.assert_close(damping.reshape(-1, h, w), 0.2 * eta_gpu + 1e-7, rtol=1e-6, atol=1e-12)
        # the stage on the GPU's own inputs
        st = before
        stage = ostep.video_ba(st, _np(target).reshape(-1, h, w, 2), _np(weight).reshape(-1, h, w, 2),
                               _np(damping).reshape(-1, h, w), ii.cpu().numpy(), jj.cpu().numpy(), t0, t1, 2,
                               1e-4, 0.1, False, opt, cfg)
        assert stage == opt, "stage 2 must not fall back in this configuration"
        if opt == "pose_depth":
            assert st["ba_info"]["failed"] == 0
        np.testing.assert_allclose(_np(video.poses), st["poses"], atol=1e-4, err_msg=f"poses after {opt}")
        np.testing.assert_allclose(_np(video.disps), st["disps"], atol=2e-4, err_msg=f"disps after {opt}")
        if opt == "depth_scale":
            n = video.counter.value
            vm = video.valid_depth_mask_small[:n].cpu().numpy()
            assert (vm != st["valid_small"][:n]).mean() < 2e-3         # fp32 threshold borderline pixels
            np.testing.assert_allclose(_np(video.depth_scale)[:n], st["depth_scale"][:n], rtol=2e-3, atol=2e-4)
            np.testing.assert_allclose(_np(video.depth_shift)[:n], st["depth_shift"][:n], rtol=2e-3, atol=2e-4)
            assert np.array_equal(_np(video.poses), before["poses"])  # stage 2 keeps the poses
        up_ref = ogeom.cvx_upsample(st["disps"][uniq.cpu().numpy()], upmask.float().cpu().numpy().reshape(-1, 576, h, w)
                                    .astype(np.float16), np.float16)
        np.testing.assert_allclose(_np(video.disps_up)[uniq.cpu().numpy()], up_ref, rtol=1e-3, atol=3e-4)


@pytest.mark.parametrize("K,h,w", [(6, 30, 40), (8, 60, 80)])
def test_update_end_to_end_matches_oracle_composition(gpu, K, h, w):
    g, video, graph = _setup(gpu, K, h, w)
    st, gr = _video_state(video), _graph_state(video, graph)
    fn = _fp32_update_fn()
    cfg = _cfg(video)
    for it, opt in enumerate(["pose_depth", "depth_scale"]):
        graph.update(t0=1, t1=K, itrs=2, EP=1e-7, opt_type=opt)
        torch.cuda.synchronize()
        out = ostep.update_step(st, gr, fn, t0=1, t1=K, itrs=2, EP=1e-7, opt_type=opt, cfg=cfg)
        assert out["stage"] == opt
        msg = f"step {it} ({opt})"
        # flows / confidences / recurrent state: fp16 evaluation of the operator
        np.testing.assert_allclose(_np(graph.target[0]), gr["target"], rtol=2e-2, atol=2e-2, err_msg=msg + " target")
        np.testing.assert_allclose(_np(graph.weight[0]), gr["weight"], rtol=2e-2, atol=1e-2, err_msg=msg + " weight")
        np.testing.assert_allclose(_np(graph.net[0]), gr["net"], rtol=2e-2, atol=2e-2, err_msg=msg + " net")
        uq = np.unique(gr["ii"])
        np.testing.assert_allclose(_np(graph.damping)[uq], gr["damping"][uq], rtol=2e-2, atol=2e-5,
                                   err_msg=msg + " eta")
        np.testing.assert_allclose(_np(video.poses), st["poses"], atol=1e-4, err_msg=msg + " poses")
        np.testing.assert_allclose(_np(video.disps), st["disps"], atol=5e-3, err_msg=msg + " disps")
        np.testing.assert_allclose(_np(video.disps_up)[uq], st["disps_up"][uq], atol=5e-3, err_msg=msg + " disps_up")
        if opt == "depth_scale":
            n = st["n"]
            np.testing.assert_allclose(_np(video.depth_scale)[:n], st["depth_scale"][:n], rtol=5e-3, atol=5e-4)
            np.testing.assert_allclose(_np(video.depth_shift)[:n], st["depth_shift"][:n], rtol=5e-3, atol=5e-4)
        assert np.array_equal(graph.age.cpu().numpy(), gr["age"])


def test_update_with_inactive_factors_matches_oracle(gpu):
    """use_inactive=True (the frontend's call, frontend.py:50-53): inactive factors that still touch the
    window are prepended to the BA's edge list (factor_graph.py:237-243)"""
    K, h, w = 7, 24, 32
    g, video, graph = _setup(gpu, K, h, w)
    old = (graph.ii == 0) | (graph.jj == 0) | (graph.ii == 2)
    graph.rm_factors(old, store=True)                   # edges of frames 0 and 2 retire to the inactive set
    assert graph.ii_inac.numel() > 0
    cfg = _cfg(video)
    before = _video_state(video)
    graph.update(t0=3, t1=K, itrs=2, use_inactive=True, opt_type="pose_depth")
    torch.cuda.synchronize()
    target, weight, damping, ii, jj, uniq, upmask, t0, t1 = graph._ba_args
    m = ((graph.ii_inac >= 0) & (graph.jj_inac >= 0)).cpu().numpy()        # t0 - 3 = 0: all of them
    want_ii = np.concatenate([graph.ii_inac.cpu().numpy()[m], graph.ii.cpu().numpy()])
    assert np.array_equal(ii.cpu().numpy(), want_ii)
    n_in = int(m.sum())
    assert torch.equal(target[:, :n_in], graph.target_inac) and torch.equal(target[:, n_in:], graph.target)
    uq_all = torch.unique(ii)
    torch.testing.assert_close(damping, 0.2 * graph.damping[uq_all] + 1e-7, rtol=1e-6, atol=1e-12)
    st = before
    ostep.video_ba(st, _np(target)[0], _np(weight)[0], _np(damping), ii.cpu().numpy(), jj.cpu().numpy(), 3, K, 2,
                   1e-4, 0.1, False, "pose_depth", cfg)
    np.testing.assert_allclose(_np(video.poses), st["poses"], atol=1e-4)
    np.testing.assert_allclose(_np(video.disps), st["disps"], atol=2e-4)
    assert np.array_equal(_np(video.poses)[:3], before["poses"][:3])
