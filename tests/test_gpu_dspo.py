"""Parity of DSPO stage 2 (disparity + scale/shift) with the dense oracle restatement."""
import numpy as np
import pytest
import torch

import glorie_slam_amd.synth as synth
from oracle import dspo as odspo, geom as ogeom

pytestmark = pytest.mark.gpu


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def problem(K=6, h=16, w=20, seed=3):
    g = synth.keyframe_graph(K=K, h=h, w=w, radius=2, seed=seed)
    rng = np.random.default_rng(seed)
    coords, _ = ogeom.reproject(g["poses"], g["disps"], g["intrinsics"], g["ii"], g["jj"])
    g["target"] = (coords + g["noise"].transpose(0, 2, 3, 1)).astype(np.float32)
    g["weight_hw2"] = np.ascontiguousarray(g["weight"].transpose(0, 2, 3, 1))
    s = rng.uniform(0.5, 2.0, K).astype(np.float32)
    q = rng.uniform(-0.05, 0.05, K).astype(np.float32)
    mono = ((g["disps"] - q[:, None, None]) / s[:, None, None] * (1 + 0.02 * rng.standard_normal(g["disps"].shape)))
    mono[rng.uniform(size=mono.shape) < 0.1] = 0.0
    g["mono"] = mono.astype(np.float32)
    g["scales"] = (s * rng.uniform(0.9, 1.1, K)).astype(np.float32)
    g["shifts"] = (q + rng.uniform(-0.01, 0.01, K)).astype(np.float32)
    g["vmask"] = rng.uniform(size=g["disps"].shape) < 0.7
    return g


def run_gpu(g, dev, itrs, edge_on=None, lm=1e-4, ep=0.1, eta=None):
    from glorie_slam_amd import _lib as L
    ctx = L.default_context()
    poses, disps = _t(g["poses"], dev), _t(g["disps"], dev)
    sc, sh = _t(g["scales"], dev), _t(g["shifts"], dev)
    eta_t = _t(g["eta"] if eta is None else eta, dev)
    K, h, w = g["K"], g["h"], g["w"]
    args = [_t(g["intrinsics"], dev), _t(g["mono"], dev), _t(g["vmask"].astype(np.uint8), dev),
            _t(g["target"], dev), _t(g["weight_hw2"], dev), _t(g["ii"], dev), _t(g["jj"], dev)]
    eo = _t(edge_on.astype(np.uint8), dev) if edge_on is not None else None
    L.check(L.load().glorie_dspo_scale_shift(
        ctx.handle, L.ptr(poses), L.ptr(disps), L.ptr(args[0]), L.ptr(args[1]), L.ptr(sc), L.ptr(sh),
        L.ptr(args[2]), L.ptr(args[3]), L.ptr(args[4]), L.ptr(eta_t), L.ptr(args[5]), L.ptr(args[6]),
        L.ptr(eo), K, len(g["ii"]), eta_t.shape[0], h, w, itrs, lm, ep, 0.01, None, L.stream_ptr()), "dspo")
    torch.cuda.synchronize()
    return disps.cpu().numpy(), sc.cpu().numpy(), sh.cpu().numpy(), ctx.ba_status()


@pytest.mark.parametrize("itrs", [1, 2])
def test_stage2_matches_dense_oracle(gpu, itrs):
    g = problem()
    d, s, q = g["disps"], g["scales"], g["shifts"]
    for _ in range(itrs):
        d, s, q, _ = odspo.ba_with_scale_shift(g["target"], g["weight_hw2"], g["eta"], g["poses"], d,
                                               g["intrinsics"], g["ii"], g["jj"], g["mono"], s, q, g["vmask"])
    gd, gs, gq, st = run_gpu(g, gpu, itrs)
    assert st[0] == 0
    np.testing.assert_allclose(gs, s, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gq, q, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gd, d, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("K,h,w", [(40, 48, 64), (64, 30, 40)])
def test_stage2_matches_dense_oracle_at_long_graph_shapes(gpu, K, h, w):
    """the same comparison at the shapes of BASELINE config 5 (TUM 384x512 -> 48x64, M = 40 depth frames) and config 4
    (ScanNet 240x320 -> 30x40, M = 64): BA_with_scale_shift (src/geom/ba.py:127-216) in the reference's dense M x M
    formulation (oracle/dspo.py: E is M x M x 2 x HW, 39 MB at M = 40) against the block-diagonal HIP solve, two calls.
    At these sizes the reduced 2 x 2 systems are a 1e-3 cancellation (H - E Q E^T over 3072 pixels) and the reference's OWN
    fp32 evaluation is only good to ~1e-4 on the shifts (measured: the fp32 oracle against the same algebra in double,
    1.2e-4 at 48x64 - and the first version of this test, which held the HIP path to 2e-5 against the fp32 oracle, failed by
    exactly that much).  So the yardstick is the double evaluation: the HIP path (fp32 per pixel, fp64 across workgroups and
    in the solve) must be at least as close to it as the reference's fp32 formulation is, and within the stated tolerance
    of the fp32 oracle widened by that rounding."""
    g = problem(K=K, h=h, w=w, seed=5)
    assert len(set(g["ii"].tolist())) == K
    res = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        d, s, q = g["disps"], g["scales"], g["shifts"]
        for _ in range(2):
            d, s, q, _ = odspo.ba_with_scale_shift(g["target"], g["weight_hw2"], g["eta"], g["poses"], d, g["intrinsics"],
                                                   g["ii"], g["jj"], g["mono"], s, q, g["vmask"], dtype=dt)
        res[tag] = [np.asarray(x, np.float64) for x in (d, s, q)]
    gd, gs, gq, st = run_gpu(g, gpu, 2)
    assert st[0] == 0
    for name, got, r32, r64, rtol, atol in (("scales", gs, res["f32"][1], res["f64"][1], 2e-4, 2e-5),
                                            ("shifts", gq, res["f32"][2], res["f64"][2], 2e-4, 2e-5),
                                            ("disps", gd, res["f32"][0], res["f64"][0], 1e-4, 2e-5)):
        e_ref = float(np.abs(r32 - r64).max())            # rounding of the reference's fp32 formulation
        e_hip = float(np.abs(got - r64).max())
        assert e_hip <= max(1.5 * e_ref, atol), (name, e_hip, e_ref)
        np.testing.assert_allclose(got, r32, rtol=rtol, atol=atol + 2.0 * e_ref, err_msg=name)


def test_stage2_edge_mask_equals_filtered_graph(gpu):
    """edge_on mask == physically removing the edges (and the eta rows of emptied frames)"""
    g = problem(K=6)
    bad = 2
    keep = (g["ii"] != bad) & (g["jj"] != bad)
    g2 = dict(g)
    for k in ("ii", "jj", "target", "weight_hw2"):
        g2[k] = g[k][keep]
    kx_all = sorted(set(g["ii"].tolist()))
    kx_f = sorted(set(g2["ii"].tolist()))
    eta_f = g["eta"][[kx_all.index(f) for f in kx_f]]
    d, s, q, _ = odspo.ba_with_scale_shift(g2["target"], g2["weight_hw2"], eta_f, g["poses"], g["disps"],
                                           g["intrinsics"], g2["ii"], g2["jj"], g["mono"], g["scales"],
                                           g["shifts"], g["vmask"])
    gd, gs, gq, st = run_gpu(g, gpu, 1, edge_on=keep)
    np.testing.assert_allclose(gd, d, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(gs, s, rtol=2e-4, atol=2e-5)
    assert np.array_equal(gd[bad], g["disps"][bad]) and gs[bad] == g["scales"][bad]


def test_stage2_failure_zeroes_all_scale_updates(gpu):
    g = problem()
    gd, gs, gq, st = run_gpu(g, gpu, 1, lm=-3.0, ep=-1.0)
    assert st[0] & 4
    assert np.array_equal(gs, g["scales"]) and np.array_equal(gq, g["shifts"])
    assert not np.array_equal(gd, g["disps"])       # dz = Q w is still applied
    d, s, q, _ = odspo.ba_with_scale_shift(g["target"], g["weight_hw2"], g["eta"], g["poses"], g["disps"],
                                           g["intrinsics"], g["ii"], g["jj"], g["mono"], g["scales"],
                                           g["shifts"], g["vmask"], lm=-3.0, ep=-1.0)
    np.testing.assert_allclose(gd, d, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("mv_thresh,mono_thres", [(0.25, 0.1), (0.05, 0.1), (0.25, 0.0)])
def test_fused_prepare_matches_reference_formulation(gpu, mv_thresh, mono_thres):
    """glorie_dspo_prepare (validity mask + nanmedian + least squares + edge filter, 4 launches)
    against the reference's op-by-op torch formulation on the same video state"""
    from test_gpu_graph import make_video
    from glorie_slam_amd import dspo as gdspo
    res = []
    for fused in (False, True):
        g, video = make_video(gpu, 7, 24, 32)
        video.cfg["tracking"]["multiview_filter"]["thresh"] = mv_thresh
        video.mono_thres = mono_thres
        # make two frames "bad": a frame whose mono prior is anti-correlated, and one with holes
        video.mono_disps[2] = -video.mono_disps[2] + 1.0
        video.mono_disps[4] += torch.linspace(0, 3, 24 * 32, device=gpu).view(24, 32).sin() * 0.3
        ii = torch.as_tensor(g["ii"], device=gpu)
        jj = torch.as_tensor(g["jj"], device=gpu)
        n = video.counter.value
        if fused:
            from glorie_slam_amd import droid_backends as db
            mv = video.cfg["tracking"]["multiview_filter"]
            eo, any_on = db.dspo_prepare(video.poses, video.disps, video.intrinsics[0].contiguous(),
                                         video.mono_disps, n, mv["thresh"], mv["visible_num"], video.mono_thres,
                                         ii, jj, video.valid_depth_mask_small, video.depth_scale, video.depth_shift)
            eo = eo.bool()
        else:
            eo, any_on = gdspo._prepare_torch(video, n, ii, jj)
            if eo is None:
                eo, any_on = torch.ones_like(ii, dtype=torch.bool), torch.ones(1, dtype=torch.int32, device=gpu)
        res.append((video.valid_depth_mask_small[:n].clone(), video.depth_scale[:n].clone(),
                    video.depth_shift[:n].clone(), eo.clone(), int(any_on.item())))
    (m0, s0, t0, e0, a0), (m1, s1, t1, e1, a1) = res
    assert float((m0 != m1).float().mean()) < 2e-3          # borderline pixels of the fp32 threshold only
    assert 0.05 < float(m0.float().mean()) < 1.0
    ok = torch.isfinite(s0)
    assert torch.equal(ok, torch.isfinite(s1))
    torch.testing.assert_close(s1[ok], s0[ok], rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(t1[ok], t0[ok], rtol=2e-3, atol=2e-4)
    assert torch.equal(e0, e1) and a0 == a1
    if mono_thres and mv_thresh == 0.25:
        assert not bool(e0.all()) and bool(e0.any())          # the filter removed some edges, not all
