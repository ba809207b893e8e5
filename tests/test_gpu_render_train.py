"""Training path of the renderer (csrc/train.hip): gradients of a depth + colour loss with respect to the feature
tables and every decoder parameter against torch autograd over the plain modules (the formulation the reference's
mapper differentiates, mapper.py:390-515), against the gradient fixture minted from the REFERENCE decoders
(tests/golden/render_grad.npz), and the Adam step against torch.optim.Adam."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _loss(depth, color, gt_depth, gt_color, sel=None):
    """mapper.py:497-505: L1 on depth + w_color * L1 on colour"""
    if sel is not None:
        return torch.abs(gt_depth - depth)[sel].sum() + 0.5 * torch.abs(gt_color - color)[sel].sum()
    return torch.abs(gt_depth - depth).sum() + 0.5 * torch.abs(gt_color - color).sum()


def _scene(gpu):
    from test_gpu_render import _reference_render_setup
    f, npc, dec, ren, ro, rd, c2w, depth, depth_zero, radius = _reference_render_setup(gpu)
    g = torch.Generator().manual_seed(3)
    gt_color = torch.rand(ro.shape[0], 3, generator=g).to(gpu)
    gt_depth = depth * (1.0 + 0.02 * torch.randn(depth.shape[0], generator=g).to(gpu))
    return f, npc, dec, ren, ro, rd, depth, radius, gt_depth, gt_color


def _run(ren, npc, dec, rd, ro, depth, radius, gt_depth, gt_color, stage, train_path, seen_only=False):
    geo = npc.geo_feats.detach().clone().requires_grad_(True)
    col = npc.col_feats.detach().clone().requires_grad_(True)
    for p in dec.parameters():
        p.grad = None
    ren.use_train_path = train_path
    d, u, c, vm, cnt = ren.render_batch_ray(npc, dec, rd, ro, rd.device, stage, gt_depth=depth, npc_geo_feats=geo,
                                            npc_col_feats=col, cloud_pos=npc.cloud_pos(), dynamic_r_query=radius)
    loss = _loss(d, c, gt_depth, gt_color, (cnt > 0) if seen_only else None)
    loss.backward()
    grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in dec.named_parameters()}
    return dict(depth=d.detach(), color=c.detach(), unc=u.detach(), mask=vm, count=cnt, loss=float(loss),
                geo=geo.grad, col=col.grad, params=grads)


@pytest.mark.parametrize("stage", ["color", "geometry"])
def test_train_path_gradients_match_torch_autograd(gpu, stage):
    f, npc, dec, ren, ro, rd, depth, radius, gt_depth, gt_color = _scene(gpu)
    dec.use_fused = False                     # the autograd reference: plain nn.Linear modules + torch ops
    a = _run(ren, npc, dec, rd, ro, depth, radius, gt_depth, gt_color, stage, train_path=False)
    b = _run(ren, npc, dec, rd, ro, depth, radius, gt_depth, gt_color, stage, train_path=True)
    torch.testing.assert_close(b["depth"], a["depth"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(b["color"], a["color"], rtol=1e-4, atol=1e-5)
    assert torch.equal(a["mask"], b["mask"]) and torch.equal(a["count"], b["count"])
    assert abs(a["loss"] - b["loss"]) < 1e-4 * abs(a["loss"])

    def close(x, y, name):
        scale = float(y.abs().max())
        assert scale > 0, name
        err = float((x - y).abs().max())
        assert err <= 2e-3 * scale + 1e-7, f"{name}: max |diff| {err:.3e} at scale {scale:.3e}"

    close(b["geo"], a["geo"], "d loss / d geo_feats")
    assert int((a["geo"].abs().sum(1) > 0).sum()) > 500
    if stage == "color":
        close(b["col"], a["col"], "d loss / d col_feats")
    used = 0
    for n, ga in a["params"].items():
        gb = b["params"][n]
        if ga is None or float(ga.abs().max()) == 0.0:
            assert gb is None or float(gb.abs().max()) < 1e-6, n        # parameters the stage does not touch
            continue
        close(gb, ga, n)
        used += 1
    assert used >= (40 if stage == "color" else 20)


def test_train_path_matches_reference_gradient_fixture(gpu):
    """gradients of the same loss computed by autograd over the REFERENCE's decoder + compositing on CPU
    (tests/golden/make_golden.py::make_render_grad)"""
    f = np.load(os.path.join(GOLD, "render_grad.npz"))
    _, npc, dec, ren, ro, rd, depth, radius, _, _ = _scene(gpu)
    gt_depth = torch.from_numpy(f["gt_depth"]).to(gpu)
    gt_color = torch.from_numpy(f["gt_color"]).to(gpu)
    # the fixture's loss leaves out rays none of whose samples has neighbours: the reference decodes those from
    # RANDOM placeholder features (decoder.py:170-171,386-387)
    b = _run(ren, npc, dec, rd, ro, depth, radius, gt_depth, gt_color, "color", train_path=True, seen_only=True)
    assert abs(b["loss"] - float(f["loss"])) < 2e-4 * abs(float(f["loss"]))

    def close(x, ref, name):
        ref = torch.from_numpy(ref).to(gpu)
        scale = float(ref.abs().max())
        err = float((x - ref).abs().max())
        assert err <= 5e-3 * scale + 1e-7, f"{name}: max |diff| {err:.3e} at scale {scale:.3e}"

    close(b["geo"], f["d_geo"], "geo_feats")
    close(b["col"], f["d_col"], "col_feats")
    n = 0
    for k in f.files:
        if k.startswith("g__"):
            close(b["params"][k[3:]], f[k], k[3:])
            n += 1
    assert n >= 40


def test_adam_step_invalidates_the_packed_inference_parameters(gpu):
    """FeatureAdam updates the parameters through raw pointers; the inference kernels read a packed copy that POINT._packed
    caches by tensor version: a step must make it stale (the mapper trains the decoders, render_img then reads them)"""
    from glorie_slam_amd import point_ops
    from glorie_slam_amd.render_train import FeatureAdam
    _, npc, dec, ren, ro, rd, depth, radius, _, _ = _scene(gpu)
    before = dec._packed().clone()
    params = list(dec.parameters())
    opt = FeatureAdam([{"params": params, "lr": 1e-2}])
    for p in params:
        p.grad = torch.ones_like(p)
    opt.step()
    after = dec._packed()
    assert not torch.equal(before, after)
    assert torch.equal(after, point_ops.pack_decoders(dec))


def test_feature_adam_matches_torch_adam(gpu):
    from glorie_slam_amd.render_train import FeatureAdam
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(500, 32, generator=g).to(gpu)
    pa, pb = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    oa = torch.optim.Adam([{"params": [pa], "lr": 3e-3}])
    ob = FeatureAdam([{"params": [pb], "lr": 3e-3}])
    for it in range(5):
        grad = torch.randn(500, 32, generator=g).to(gpu) * (0.1 + it)
        pa.grad, pb.grad = grad.clone(), grad.clone()
        oa.step()
        ob.step()
        torch.testing.assert_close(pb.detach(), pa.detach(), rtol=1e-5, atol=1e-7)
    # row mask: rows outside the frustum keep value and moments
    mask = torch.zeros(500, dtype=torch.bool, device=gpu)
    mask[::3] = True
    before = pb.detach().clone()
    pb.grad = torch.ones_like(pb)
    ob.step(row_masks={id(pb): mask})
    assert torch.equal(pb.detach()[~mask], before[~mask]) and not torch.equal(pb.detach()[mask], before[mask])


def test_feature_adam_follows_lr_changes_between_steps(gpu):
    """the mapper rewrites optimizer.param_groups[i]['lr'] every step (mapper.py:412-414: geometry stage -> colour stage);
    the multi-tensor launch keeps a device table of (pointers, lr, betas, eps) per small tensor, and the gradient buffers
    usually come back at the same address - the table must be rebuilt when only the hyper-parameters changed"""
    from glorie_slam_amd.render_train import FeatureAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 32), (32,), (64, 93), (1, 7)]                    # small tensors: the glorie_adam_multi path
    p0 = [torch.randn(*s, generator=g).to(gpu) for s in shapes]
    pa = [p.clone().requires_grad_(True) for p in p0]
    pb = [p.clone().requires_grad_(True) for p in p0]
    oa = torch.optim.Adam([{"params": pa[:2], "lr": 1e-3}, {"params": pa[2:], "lr": 3e-2}])
    ob = FeatureAdam([{"params": pb[:2], "lr": 1e-3}, {"params": pb[2:], "lr": 3e-2}])
    grads = [torch.empty_like(p) for p in pb]                        # ONE gradient buffer per tensor for the whole loop
    for it in range(6):
        if it == 2:                                                  # stage switch: both groups change their rate
            for o in (oa, ob):
                o.param_groups[0]["lr"], o.param_groups[1]["lr"] = 5e-3, 5e-3
        if it == 4:
            for o in (oa, ob):
                o.param_groups[1]["lr"] = 1e-4
        for a, b, gb in zip(pa, pb, grads):
            gb.copy_(torch.randn(a.shape, generator=g).to(gpu) * (0.2 + it))
            a.grad, b.grad = gb.clone(), gb
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            # (a stale learning rate is an O(lr) = 1e-3 .. 3e-2 error per step; fp32 rounding of the update is ~1e-7)
            torch.testing.assert_close(b.detach(), a.detach(), rtol=1e-5, atol=1e-6)
