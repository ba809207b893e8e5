"""HIP path against fixtures minted from the reference's own Python geometry (tests/golden/make_pins.py:mint_geometry:
projective_ops.projective_transform, ba.BA_with_scale_shift, ba.BA executed with a lietorch stand-in) and the mirrors
that carry the reference's names for rows A6 / A7 / B8 / B11 (glorie_slam_amd.projective_ops, .ba, droid_net.cvx_upsample)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _g(name):
    return np.load(os.path.join(HERE, "golden", name))


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_projective_transform_mirror_equals_reference(gpu):
    from glorie_slam_amd import projective_ops as pops
    from glorie_slam_amd.lie import SE3
    P = _g("pops.npz")
    poses = SE3(_t(P["poses"], gpu)[None])
    args = (poses, _t(P["disps"], gpu)[None], _t(P["intr"], gpu)[None], _t(P["ii"], gpu), _t(P["jj"], gpu))
    c, v = pops.projective_transform(*args)
    np.testing.assert_allclose(c.cpu().numpy(), P["coords"], rtol=0, atol=2e-5)
    np.testing.assert_array_equal(v.cpu().numpy(), P["valid"])
    c2, v2, (Ji, Jj, Jz) = pops.projective_transform(*args, jacobian=True)
    assert torch.equal(c2, c) and torch.equal(v2, v)
    for name, got in (("Ji", Ji), ("Jj", Jj), ("Jz", Jz)):
        want = P[name]
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-5 * np.abs(want).max())
    g0 = pops.coords_grid(6, 8, gpu)
    assert g0.shape == (6, 8, 2) and float(g0[2, 5, 0]) == 5.0 and float(g0[2, 5, 1]) == 2.0
    pts, J = pops.iproj(args[1], args[2], jacobian=True)
    assert pts.shape == (1, 5, 6, 8, 4) and float(J[..., 3].min()) == 1.0 and float(J[..., :3].abs().max()) == 0.0


def test_dspo_stage2_kernel_equals_reference_BA_with_scale_shift(gpu):
    """glorie_dspo_scale_shift behind the reference's name and argument list against two sequential calls of the
    reference's ba.BA_with_scale_shift (ba.py:127-216): disparities, scales, shifts"""
    from glorie_slam_amd.ba import BA_with_scale_shift
    from glorie_slam_amd.lie import SE3
    P, f = _g("pops.npz"), _g("ba_scale_shift.npz")
    poses = SE3(_t(P["poses"], gpu)[None])
    disps = _t(P["disps"], gpu)[None]
    scales, shifts = _t(P["scales0"], gpu), _t(P["shifts0"], gpu)
    d0 = disps.clone()
    for it in range(2):
        poses, disps, wqs = BA_with_scale_shift(_t(P["target"], gpu), _t(P["weight"], gpu), _t(f["eta_rows"], gpu), poses,
                                                disps, _t(P["intr"], gpu)[None], _t(P["ii"], gpu), _t(P["jj"], gpu),
                                                _t(P["mono"], gpu)[None], scales[None], shifts[None],
                                                _t(P["vmask"], gpu)[None], 0, 1e-4, 0.1, alpha=0.01)
        scales, shifts = wqs[0, :, 0], wqs[0, :, 1]
        np.testing.assert_allclose(disps[0].cpu().numpy(), f[f"disps_{it}"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(wqs[0].cpu().numpy(), f[f"wqs_{it}"], rtol=2e-4, atol=2e-5)
    assert torch.equal(d0, _t(P["disps"], gpu)[None])                       # inputs untouched


def test_ba_kernel_agrees_with_reference_python_BA(gpu):
    """glorie_ba (one Gauss-Newton iteration) against the reference's Python BA (ba.py:34-121), same tolerances and the
    same documented differences as the oracle's twin (tests/test_pins.py::test_oracle_stage1_agrees_with_reference_python_BA)"""
    import glorie_slam_amd.droid_backends as db
    P, f = _g("pops.npz"), _g("ba_python.npz")
    eta = _g("ba_scale_shift.npz")["eta_rows"]
    keep = P["ii"] != P["jj"]
    K = P["poses"].shape[0]
    poses, disps = _t(P["poses"], gpu), _t(P["disps"], gpu)
    tgt = _t(P["target"][0][keep].transpose(0, 3, 1, 2), gpu)
    wgt = _t(P["weight"][0][keep].transpose(0, 3, 1, 2), gpu)
    db.ba(poses, disps, _t(P["intr"][0], gpu), torch.zeros_like(disps), tgt, wgt, _t(eta, gpu), _t(P["ii"][keep], gpu),
          _t(P["jj"][keep], gpu), 1, K, 1, 1e-4, 0.1, False, False)
    p, d = poses.cpu().numpy(), disps.cpu().numpy()
    step_t = np.abs(f["poses"][:, :3] - P["poses"][:, :3]).max()
    step_q = np.abs(f["poses"][:, 3:] - P["poses"][:, 3:]).max()
    step_d = np.abs(f["disps"] - P["disps"]).max()
    assert np.abs(p[:, :3] - f["poses"][:, :3]).max() < 2e-3 * step_t
    assert np.abs(p[:, 3:] - f["poses"][:, 3:]).max() < 2e-3 * step_q
    assert np.abs(d - f["disps"]).max() < 3e-2 * step_d


def test_cvx_upsample_mirror_equals_reference(gpu):
    from glorie_slam_amd.droid_net import cvx_upsample, upsample_disp
    f = _g("cvx_upsample.npz")
    up = cvx_upsample(_t(f["data"], gpu), _t(f["mask"], gpu))
    np.testing.assert_allclose(up.cpu().numpy(), f["up"], rtol=1e-5, atol=1e-6)
    up2 = upsample_disp(_t(f["data"][..., 0], gpu)[None], _t(f["mask"], gpu)[None])
    assert up2.shape == (1, 2, 40, 56) and torch.equal(up2[0], up[..., 0])
    two = torch.cat([_t(f["data"], gpu), 2 * _t(f["data"], gpu)], -1)           # dim = 2 field
    up3 = cvx_upsample(two, _t(f["mask"], gpu))
    assert torch.equal(up3[..., 0], up[..., 0]) and torch.allclose(up3[..., 1], 2 * up[..., 0], rtol=1e-6)


def test_get_scale(gpu):
    from glorie_slam_amd.neural_point import get_scale
    a = torch.rand(1000, device=gpu) + 0.5
    assert abs(float(get_scale(a, 1.7 * a)) - 1.7) < 1e-5
