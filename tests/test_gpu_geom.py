"""Parity of the HIP geometry kernels with the oracle."""
import numpy as np
import pytest
import torch

import glorie_slam_amd.synth as synth
from oracle import geom as ogeom

pytestmark = pytest.mark.gpu


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.fixture(scope="module")
def graph():
    return synth.keyframe_graph(K=6, h=30, w=40, radius=3)


def test_reproject(gpu, graph):
    from glorie_slam_amd import droid_backends as db
    g = graph
    ii = np.concatenate([g["ii"], [2]]).astype(np.int64)   # + one stereo edge (ii == jj)
    jj = np.concatenate([g["jj"], [2]]).astype(np.int64)
    ref_c, ref_v = ogeom.reproject(g["poses"], g["disps"], g["intrinsics"], ii, jj)
    c, v = db.reproject(_t(g["poses"], gpu), _t(g["disps"], gpu), _t(g["intrinsics"], gpu),
                        _t(ii, gpu), _t(jj, gpu))
    # strict fp32 (no contraction) in the reference's operation order: bit-identical to the oracle
    assert np.array_equal(c.cpu().numpy(), ref_c)
    assert np.array_equal(v.cpu().numpy(), ref_v)


def test_frame_distance(gpu, graph):
    from glorie_slam_amd import droid_backends as db
    g = graph
    K = g["K"]
    ii, jj = np.meshgrid(np.arange(K), np.arange(K), indexing="ij")
    ii, jj = ii.reshape(-1).astype(np.int64), jj.reshape(-1).astype(np.int64)
    for beta in (0.3, 0.75):
        ref = ogeom.frame_distance(g["poses"], g["disps"], g["intrinsics"][0], ii, jj, beta)
        got = db.frame_distance(_t(g["poses"], gpu), _t(g["disps"], gpu), _t(g["intrinsics"][0], gpu),
                                _t(ii, gpu), _t(jj, gpu), beta).cpu().numpy()
        # same per-thread accumulation order, same tree, no contraction: the distances graph topology is
        # thresholded on are bit-identical to the oracle
        assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_frame_distance_invalid_returns_1000(gpu, graph):
    from glorie_slam_amd import droid_backends as db
    g = graph
    poses = g["poses"].copy()
    poses[1, :3] = (0, 0, -50.0)  # everything lands behind the camera
    ii = np.array([0], np.int64)
    jj = np.array([1], np.int64)
    got = db.frame_distance(_t(poses, gpu), _t(g["disps"], gpu), _t(g["intrinsics"][0], gpu),
                            _t(ii, gpu), _t(jj, gpu), 0.3).cpu().numpy()
    assert got[0] == 1000.0


def test_iproj(gpu, graph):
    from glorie_slam_amd import droid_backends as db
    g = graph
    ref = ogeom.iproj(g["poses"], g["disps"], g["intrinsics"][0])
    got = db.iproj(_t(g["poses"], gpu), _t(g["disps"], gpu), _t(g["intrinsics"][0], gpu)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_depth_filter(gpu):
    from glorie_slam_amd import droid_backends as db
    g = synth.keyframe_graph(K=9, h=24, w=32, radius=2)
    ix = np.array([0, 3, 4, 8], np.int64)
    thresh = (0.01 * (1.0 / g["disps"][ix]).mean((1, 2))).astype(np.float32) * 4
    ref = ogeom.depth_filter(g["poses"], g["disps"], g["intrinsics"][0], ix, thresh)
    got = db.depth_filter(_t(g["poses"], gpu), _t(g["disps"], gpu), _t(g["intrinsics"][0], gpu),
                          _t(ix, gpu), _t(thresh, gpu)).cpu().numpy()
    assert ref.max() >= 2 and (ref > 0).mean() > 0.2
    # integer counts: bit-exact (the kernels of geom.hip round every fp32 operation on its own, like the oracle)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("half", [True, False])
def test_cvx_upsample(gpu, half):
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(0)
    B, h, w = 5, 12, 18
    disps = rng.uniform(0.2, 1.0, (B, h, w)).astype(np.float32)
    ix = np.array([3, 0, 4], np.int64)
    mask = (rng.standard_normal((3, 576, h, w)) * 2).astype(np.float16 if half else np.float32)
    ref = ogeom.cvx_upsample(disps[ix], mask, np.float16 if half else None)
    up = torch.zeros(B, 8 * h, 8 * w, device=gpu)
    db.cvx_upsample(_t(disps, gpu), _t(ix, gpu), _t(mask, gpu), up, softmax_f32=False)
    got = up.cpu().numpy()
    np.testing.assert_allclose(got[ix], ref, rtol=1e-3 if half else 1e-5, atol=2e-4 if half else 1e-6)
    assert np.all(got[[1, 2]] == 0)
    if half:
        # channels-last logits (layout of the upmask convolution) take the nhwc kernel: same values
        up2 = torch.zeros(B, 8 * h, 8 * w, device=gpu)
        mcl = _t(mask, gpu).contiguous(memory_format=torch.channels_last)
        db.cvx_upsample(_t(disps, gpu), _t(ix, gpu), mcl, up2, softmax_f32=False)
        assert torch.equal(up2, up)
