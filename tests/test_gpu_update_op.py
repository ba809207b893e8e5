"""Fused update operator (csrc/gru.hip + droid_net.FusedUpdate) against the plain fp32 PyTorch
module with the same weights (the tier's floating-point reference for this operator).

Tolerance: the reference runs this operator under fp16 autocast (factor_graph.py:211), so the
comparison against fp32 carries the fp16 noise of 12 stacked convolutions: 2e-2 absolute on
O(1) activations, and the fused path must be at least as close to fp32 as the autocast path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl_half(n, c, h, w, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(n, c, h, w, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)


def test_bias_act_slices(gpu):
    from glorie_slam_amd import update_ops as U
    x = _cl_half(3, 64, 5, 7, gpu, 0)
    b = torch.linspace(-1, 1, 64, device=gpu)
    wide = torch.zeros(3, 448, 5, 7, device=gpu, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    U.bias_act(x, b, U.ACT_RELU, out=wide[:, 384:448])
    ref = F.relu(x.float() + b.view(1, -1, 1, 1)).half()
    assert torch.equal(wide[:, 384:448], ref)
    assert float(wide[:, :384].abs().max()) == 0.0
    y = U.bias_act(x.clone(memory_format=torch.preserve_format), b, U.ACT_SIGMOID)
    torch.testing.assert_close(y.float(), torch.sigmoid(x.float() + b.view(1, -1, 1, 1)), atol=1e-3, rtol=1e-3)
    z = U.bias_act(x.clone(memory_format=torch.preserve_format), None, U.ACT_NONE)
    assert torch.equal(z, x)
    with pytest.raises(RuntimeError):
        U.bias_act(x.contiguous(), b, U.ACT_RELU)          # NCHW rows are not channels-last


def test_gru_gates_match_torch(gpu):
    from glorie_slam_amd import update_ops as U
    n, h, w = 4, 6, 9
    net = _cl_half(n, 128, h, w, gpu, 1)
    wn = _cl_half(n, 128, h, w, gpu, 2)
    zr = _cl_half(n, 256, h, w, gpu, 3)
    qc = _cl_half(n, 128, h, w, gpu, 4)
    bw = torch.randn(128, device=gpu)
    G = torch.randn(128, 384, device=gpu) / 11.0
    Gb = torch.randn(384, device=gpu)
    g = U.gru_glo_terms(wn, bw, net, G, Gb, parts=5)
    glo = (torch.sigmoid(wn.float() + bw.view(1, -1, 1, 1)) * net.float()).mean((2, 3))
    torch.testing.assert_close(g, (glo.double() @ G.double() + Gb.double()).float(), atol=2e-5, rtol=1e-4)
    assert torch.equal(g, U.gru_glo_terms(wn, bw, net, G, Gb, parts=5))     # deterministic
    hx = torch.zeros(n, 448, h, w, device=gpu, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    z = torch.empty_like(net)
    U.gru_gate_zr(zr, g[:, :256], net, z, hx[:, :128])
    zref = torch.sigmoid(zr[:, :128].float() + g[:, :128].view(n, 128, 1, 1))
    rref = torch.sigmoid(zr[:, 128:].float() + g[:, 128:256].view(n, 128, 1, 1)) * net.float()
    torch.testing.assert_close(z.float(), zref, atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(hx[:, :128].float(), rref, atol=2e-3, rtol=2e-3)
    out = torch.empty_like(net)
    U.gru_gate_q(qc, g[:, 256:], z, net, out, out2=hx[:, :128])
    q = torch.tanh(qc.float() + g[:, 256:].reshape(n, 128, 1, 1))
    ref = (1 - z.float()) * net.float() + z.float() * q
    torch.testing.assert_close(out.float(), ref, atol=3e-3, rtol=2e-3)
    assert torch.equal(hx[:, :128], out)


def test_segment_mean(gpu):
    from glorie_slam_amd import update_ops as U
    n, h, w = 7, 5, 6
    wide = _cl_half(n, 384, h, w, gpu, 5)
    x = wide[:, 256:384]
    ix = torch.tensor([0, 2, 2, 0, 3, 2, 0], device=gpu)           # group 1 is empty -> zeros
    b = torch.randn(128, device=gpu)
    out = U.segment_mean(x, ix, 4, bias=b, relu=True)
    act = F.relu(x.float() + b.view(1, -1, 1, 1))
    for gidx in range(4):
        m = ix == gidx
        ref = act[m].mean(0) if bool(m.any()) else torch.zeros_like(act[0])
        torch.testing.assert_close(out[gidx].float(), ref, atol=2e-3, rtol=2e-3)
    plain = U.segment_mean(x, ix, 4)
    torch.testing.assert_close(plain[2].float(), x.float()[ix == 2].mean(0), atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("K,groups", [(2, 2), (1, 1), (3, 1)])
def test_conv3x3_small_matches_conv2d(gpu, K, groups):
    from glorie_slam_amd import update_ops as U
    n, h, w = 3, 7, 10                                               # 210 pixels: ragged last tile
    x = _cl_half(n, 384, h, w, gpu, 6)
    g = torch.Generator(device="cpu").manual_seed(8)
    ws = [(torch.randn(K, 128, 3, 3, generator=g) / 20).to(gpu) for _ in range(groups)]
    ob = torch.randn(groups * K, generator=g).to(gpu)
    ib = torch.randn(128 * groups, generator=g).to(gpu)
    acts = [U.ACT_NONE, U.ACT_SIGMOID, U.ACT_SOFTPLUS, U.ACT_RELU][:groups]
    out = U.conv3x3_small(x, U.pack_conv3x3_small(ws), ob, K, acts, scale=0.5, in_bias=ib, in_relu=True)
    assert tuple(out.shape) == (groups, n, h, w, K)
    for gi in range(groups):
        xin = F.relu(x[:, 128 * gi:128 * (gi + 1)].float() + ib[128 * gi:128 * (gi + 1)].view(1, -1, 1, 1))
        xin = xin.half().float()                                     # the kernel rounds its operand to fp16
        ref = F.conv2d(xin, ws[gi].half().float(), ob[K * gi:K * (gi + 1)], padding=1)
        ref = [ref, torch.sigmoid(ref), F.softplus(ref), F.relu(ref)][gi] * 0.5
        torch.testing.assert_close(out[gi], ref.permute(0, 2, 3, 1), atol=2e-3, rtol=2e-3)
    # no input transform
    out2 = U.conv3x3_small(x, U.pack_conv3x3_small(ws), None, K, [U.ACT_NONE] * groups)
    ref = F.conv2d(x[:, :128].float(), ws[0].half().float(), None, padding=1)
    torch.testing.assert_close(out2[0], ref.permute(0, 2, 3, 1), atol=2e-3, rtol=2e-3)


def _inputs(dev, n, h, w, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    net = torch.tanh(r(1, n, 128, h, w))
    inp = torch.relu(r(1, n, 128, h, w))
    corr = 2.0 * r(1, n, 196, h, w)
    flow = torch.clamp(4.0 * r(1, n, 4, h, w), -64, 64)
    return net, inp, corr, flow


@pytest.mark.parametrize("n,h,w", [(6, 12, 16), (10, 30, 40)])
def test_fused_update_matches_fp32_module(gpu, n, h, w):
    from glorie_slam_amd.droid_net import UpdateModule, HalfUpdate, FusedUpdate
    torch.manual_seed(7)
    mod = UpdateModule().to(gpu).eval()
    net, inp, corr, flow = _inputs(gpu, n, h, w)
    ii = torch.tensor([i // 3 for i in range(n)], device=gpu)
    with torch.no_grad():
        ref = mod(net, inp, corr, flow, ii, None)
    fused = FusedUpdate(mod)(net, inp, corr, flow, ii, None)
    half = HalfUpdate(mod)(net, inp, corr, flow, ii, None)
    names = ["net", "delta", "weight", "eta", "upmask"]
    for name, a, b, c in zip(names, ref, fused, half):
        assert tuple(a.shape) == tuple(b.shape), name
        err_f = float((a.float() - b.float()).abs().max())
        err_h = float((a.float() - c.float()).abs().max())
        scale = max(1.0, float(a.abs().max()))
        assert err_f <= 2e-2 * scale, (name, err_f)
        assert err_f <= 2.0 * err_h + 2e-3 * scale, (name, err_f, err_h)


def test_fused_update_on_reference_fixture(gpu):
    """FusedUpdate on the inputs of tests/golden/update_module.npz against the outputs the REFERENCE's
    UpdateModule produced for them (fixture F1, minted from /root/reference/src/modules/droid_net by
    tests/golden/make_golden.py; fp32 on CPU).  Tolerance = SURVEY 8(d) for the fp16 path: rel 2e-2 / abs 1e-2."""
    import os
    import numpy as np
    from glorie_slam_amd.droid_net import UpdateModule, FusedUpdate
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "update_module.npz"))
    torch.manual_seed(43)
    mod = UpdateModule().eval()
    psum = float(sum(p.detach().double().abs().sum() for p in mod.parameters()))
    assert abs(psum - float(f["param_abs_sum"])) < 1e-6 * psum          # the fixture's weights
    mod = mod.to(gpu)
    fi = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "update_module_inputs.npz"))
    x_net, x_inp, x_corr, x_flow = (torch.from_numpy(fi[k]).to(gpu) for k in ("net", "inp", "corr", "flow"))   # as minted
    ii = torch.from_numpy(f["ii"]).to(gpu)
    out = FusedUpdate(mod)(x_net, x_inp, x_corr, x_flow, ii, torch.from_numpy(f["jj"]).to(gpu))
    for got, key in zip(out, ("net", "delta", "weight", "eta", "upmask")):
        ref = torch.from_numpy(f[key].astype(np.float32)).to(gpu)
        assert tuple(got.shape) == tuple(ref.shape), key
        torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=1e-2 if key != "eta" else 2e-4,
                                   msg=lambda m, k=key: f"{k}: {m}")


def test_hoisted_context_term_equals_full_gate_convolution(gpu):
    """conv([net|inp|corr|flow]) == conv([net|corr|flow]) + conv_inp(inp): the gates with the context part
    evaluated once (FusedUpdate.precompute_context, fp16 per-pixel term added in the epilogue) against the
    448-channel convolution; refreshed when `inp` or a GRU weight changes"""
    from glorie_slam_amd.droid_net import UpdateModule, FusedUpdate
    torch.manual_seed(11)
    mod = UpdateModule().to(gpu).eval()
    net, inp, corr, flow = _inputs(gpu, 5, 12, 16, seed=5)
    ii = torch.tensor([0, 0, 1, 1, 2], device=gpu)
    full, hoist = FusedUpdate(mod), FusedUpdate(mod)
    full.hoist_inp = False
    a = full(net, inp, corr, flow, ii, None)
    b = hoist(net, inp, corr, flow, ii, None)
    assert hoist._pre is not None and full._pre is None
    for name, x, y in zip(["net", "delta", "weight", "eta", "upmask"], a, b):
        scale = max(1.0, float(x.float().abs().max()))
        assert float((x.float() - y.float()).abs().max()) <= 4e-3 * scale, name
    # a new context tensor and a weight update are both picked up
    inp2 = (inp.float() * 0.5 + 0.1).to(inp.dtype)
    with torch.no_grad():
        mod.gru.convz.weight.mul_(0.9)
    a = full(net, inp2, corr, flow, ii, None)
    b = hoist(net, inp2, corr, flow, ii, None)
    for name, x, y in zip(["net", "delta", "weight", "eta", "upmask"], a, b):
        scale = max(1.0, float(x.float().abs().max()))
        assert float((x.float() - y.float()).abs().max()) <= 4e-3 * scale, name


def test_fused_update_without_graph_aggregation_and_repack(gpu):
    from glorie_slam_amd.droid_net import UpdateModule, FusedUpdate
    torch.manual_seed(9)
    mod = UpdateModule().to(gpu).eval()
    net, inp, corr, flow = _inputs(gpu, 4, 8, 8, seed=3)
    fu = FusedUpdate(mod)
    out = fu(net, inp, corr, flow)
    assert len(out) == 3
    with torch.no_grad():
        ref = mod(net, inp, corr, flow)
        torch.testing.assert_close(out[1].float(), ref[1], atol=2e-2, rtol=2e-2)
        # a parameter update must be picked up (weights are re-packed on version change)
        mod.delta[2].bias.add_(1.0)
        ref2 = mod(net, inp, corr, flow)
    out2 = fu(net, inp, corr, flow)
    torch.testing.assert_close(out2[1].float(), ref2[1], atol=2e-2, rtol=2e-2)
    # the recurrent state can be fed back as returned (channels-last strides)
    out3 = fu(out2[0], inp, corr, flow)
    assert torch.isfinite(out3[0].float()).all()


@pytest.mark.parametrize("n,h,w", [(5, 12, 16), (3, 10, 13), (36, 60, 80)])
def test_global_context_terms_fused_into_the_1x1_convolution(gpu, n, h, w):
    """gate terms from the convolution epilogue's per-tile partial sums (tiles that straddle two maps, a ragged last tile)
    against the two-launch form on the stored map and against fp32 torch"""
    from glorie_slam_amd import update_ops as U
    net = _cl_half(n, 128, h, w, gpu, 61)
    g = torch.Generator(device="cpu").manual_seed(62)
    ww = (torch.randn(128, 128, 1, 1, generator=g) / 11).to(gpu)
    bw = torch.randn(128, generator=g).to(gpu)
    G = (torch.randn(128, 384, generator=g) / 11).to(gpu).contiguous()
    Gb = torch.randn(384, generator=g).to(gpu)
    wp = U.pack_conv_igemm(ww)
    fused = U.gru_glo_terms_fused(net, wp, bw, G, Gb)
    wn = U.conv_igemm(net, None, wp, 1, 128, torch.empty_like(net))
    two = U.gru_glo_terms(wn, bw, net, G, Gb)
    torch.testing.assert_close(fused, two, atol=2e-3, rtol=2e-3)      # `two` rounds the map to fp16 in between
    x = net.float()
    wmap = F.conv2d(x, ww.half().float())
    glo = (torch.sigmoid(wmap + bw.view(1, -1, 1, 1)) * x).mean((2, 3))
    torch.testing.assert_close(fused, glo @ G + Gb, atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("n,h,w", [(5, 12, 16), (3, 10, 13), (36, 60, 80)])
def test_head_taps_fused_into_the_hidden_convolution(gpu, n, h, w):
    """delta / weight heads (droid_net.py:85-93) with their hidden maps consumed in the convolution's epilogue against the
    stored-map form (conv_igemm + conv3x3_small) and against fp32 torch; the trailing (GraphAgg) channels are bit-identical"""
    from glorie_slam_amd import update_ops as U
    x = _cl_half(n, 128, h, w, gpu, 71)
    g = torch.Generator(device="cpu").manual_seed(72)
    w1 = (torch.randn(384, 128, 3, 3, generator=g) / 30).to(gpu)
    b1 = (torch.randn(384, generator=g) / 4).to(gpu).contiguous()
    w2 = [(torch.randn(2, 128, 3, 3, generator=g) / 30).to(gpu) for _ in range(2)]
    b2 = torch.randn(4, generator=g).to(gpu)
    wp = U.pack_conv_igemm(w1)
    acts = (U.ACT_NONE, U.ACT_SIGMOID)
    rest = torch.empty((n, 128, h, w), dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    rows = U.conv_igemm_heads(x, wp, 9, 384, b1, U.pack_head_taps(w2), 2, out=rest)
    fused = U.conv_stencil(rows, b2, n, h, w, 2, 2, acts)
    h1 = torch.empty((n, 384, h, w), dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    U.conv_igemm(x, None, wp, 9, 384, h1, terms=b1, act=U.ACT_RELU)
    two = U.conv3x3_small(h1, U.pack_conv3x3_small(w2), b2, 2, acts)
    assert torch.equal(rest, h1[:, 256:384])
    torch.testing.assert_close(fused, two, atol=2e-4, rtol=2e-4)        # same fp16 operands, another summation order
    hid = torch.relu(F.conv2d(x.float(), w1.half().float(), b1, padding=1)).half().float()
    for k in range(2):
        ref = F.conv2d(hid[:, 128 * k:128 * k + 128], w2[k].half().float(), b2[2 * k:2 * k + 2], padding=1)
        if k == 1:
            ref = torch.sigmoid(ref)
        torch.testing.assert_close(fused[k], ref.permute(0, 2, 3, 1), atol=3e-3, rtol=3e-3)


@pytest.mark.parametrize("m,h,w,f32", [(3, 9, 11, False), (2, 10, 13, True), (8, 60, 80, False)])
def test_upmask_convolution_with_the_convex_upsampling_as_its_epilogue(gpu, m, h, w, f32):
    """glorie_conv_upsample against conv_igemm (stored fp16 logits) + cvx_upsample: same bits; frames that are not in ix
    stay untouched"""
    from glorie_slam_amd import update_ops as U, droid_backends as db
    g = torch.Generator(device="cpu").manual_seed(81)
    x = _cl_half(m, 128, h, w, gpu, 82)
    wt = (torch.randn(576, 128, 1, 1, generator=g) / 6).to(gpu)
    bias = torch.randn(576, generator=g).to(gpu)
    B = m + 3
    disps = (torch.rand(B, h, w, generator=g) + 0.2).to(gpu)
    ix = torch.randperm(B, generator=g)[:m].sort().values.to(gpu)
    logits = torch.empty((m, 576, h, w), dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    U.conv_igemm(x, None, U.pack_conv_igemm(wt), 1, 576, logits, terms=bias.contiguous())
    ref = torch.full((B, 8 * h, 8 * w), -1.0, device=gpu)
    db.cvx_upsample(disps, ix, logits, ref, softmax_f32=f32)
    got = torch.full((B, 8 * h, 8 * w), -1.0, device=gpu)
    wp, bp = U.pack_upmask_conv(wt, bias)
    U.conv_upsample(U.LazyUpmask(x, wp, bp), disps, ix, got, softmax_f32=f32)
    assert torch.equal(got, ref)
    untouched = [f for f in range(B) if f not in set(ix.tolist())]
    assert float((got[untouched] + 1.0).abs().max()) == 0.0


def test_wide_convolutions_are_repeatable_at_full_size(gpu):
    """the LDS-DMA pieces of a K tile must all have landed before the tile is read: a wait that counts something else
    into vmcnt shows up as run-to-run differences in the upper channel half once two workgroups share a CU (1350 pixel
    tiles here).  Four runs of the z|r gate launch and of the plain 320 -> 256 layer give the same bits."""
    from glorie_slam_amd import update_ops as U
    n, h, w = 36, 60, 80
    net, wide, pre = _cl_half(n, 128, h, w, gpu, 91), _cl_half(n, 320, h, w, gpu, 92), _cl_half(n, 256, h, w, gpu, 93)
    dynx = wide[:, 128:320]
    g = torch.Generator(device="cpu").manual_seed(94)
    wzr = U.pack_conv_igemm((torch.randn(256, 320, 3, 3, generator=g) / 54).to(gpu))
    terms = torch.randn(n, 256, generator=g).to(gpu)
    bias = torch.randn(256, generator=g).to(gpu)
    ref = None
    for rep in range(4):
        z = torch.empty_like(net)
        rnet = torch.empty_like(net)
        U.conv_igemm(net, dynx, wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms, net=net, out2=rnet, pre=pre)
        o = torch.empty((n, 256, h, w), dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
        U.conv_igemm(net, dynx, wzr, 9, 256, o, terms=bias, act=U.ACT_RELU)
        torch.cuda.synchronize()
        cur = (z.clone(), rnet.clone(), o.clone())
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert torch.equal(a, b)


def test_fused_update_on_the_channels_last_lookup(gpu):
    """corr_encoder[0] as a 1x1 implicit-GEMM convolution over the channels-last lookup (permuted weight columns, bias and
    ReLU in the epilogue) against the library GEMM over the planar map: same operator, only the fp16 GEMM's summation order
    differs"""
    from glorie_slam_amd.droid_net import FusedUpdate, UpdateModule
    torch.manual_seed(6)
    mod = UpdateModule().to(gpu).eval()
    n, h, w = 5, 12, 16
    ii = torch.tensor([0, 0, 1, 2, 2], device=gpu)
    jj = torch.tensor([1, 2, 0, 0, 1], device=gpu)
    net = torch.randn(1, n, 128, h, w, device=gpu).half()
    inp = torch.randn(1, n, 128, h, w, device=gpu).half()
    corr = torch.randn(1, n, 196, h, w, device=gpu).half()
    flow = torch.randn(1, n, 4, h, w, device=gpu)
    cl = torch.zeros(n, 4, 8, 8, h, w, device=gpu, dtype=torch.float16)
    cl[:, :, :7, :7] = corr.view(n, 4, 7, 7, h, w).permute(0, 1, 3, 2, 4, 5)
    cl = cl.view(n, 256, h, w).contiguous(memory_format=torch.channels_last)
    a, b = FusedUpdate(mod), FusedUpdate(mod)
    ra = a(net, inp, corr, flow, ii, jj)
    rb = b(net, inp, cl, flow, ii, jj)
    for x, y in zip(ra, rb):
        torch.testing.assert_close(x.float(), y.float(), atol=4e-3, rtol=4e-3)


def test_shared_context_term_equals_per_edge_term(gpu, monkeypatch):
    """Edges with the same source keyframe have the same context features: the hoisted gate term kept once per
    keyframe and read through pre_map gives the bits of the per-edge term (both forms add it in the epilogue)."""
    from glorie_slam_amd.droid_net import FusedUpdate, UpdateModule
    torch.manual_seed(5)
    mod = UpdateModule().to(gpu).eval()
    n, h, w = 7, 12, 16
    ii = torch.tensor([0, 0, 1, 3, 3, 3, 4], device=gpu)
    jj = torch.tensor([1, 2, 0, 1, 2, 4, 3], device=gpu)
    table = torch.randn(6, 128, h, w, device=gpu).half()
    inp = table[ii][None]
    net = torch.randn(1, n, 128, h, w, device=gpu).half()
    corr = torch.randn(1, n, 196, h, w, device=gpu).half()
    flow = torch.randn(1, n, 4, h, w, device=gpu)
    frames, ix = torch.unique(ii, sorted=True, return_inverse=True)
    a, b = FusedUpdate(mod), FusedUpdate(mod)
    ra = a(net, inp, corr, flow, ii, jj)
    rb = b(net, inp, corr, flow, ii, jj, context=(table, frames, ix))
    assert b._pre is None and b._pre_kf is not None and tuple(b._pre_kf[:4].shape) == (4, 384, h, w)
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)
    # a changed table row is picked up (version key), an unchanged call does not recompute
    key = b._ctx_key
    b(net, inp, corr, flow, ii, jj, context=(table, frames, ix))
    assert b._ctx_key == key
    table[3] += 1.0
    rc = b(net, table[ii][None], corr, flow, ii, jj, context=(table, frames, ix))
    rd = a(net, table[ii][None].clone(), corr, flow, ii, jj)
    assert b._ctx_key != key
    for x, y in zip(rc, rd):
        assert torch.equal(x, y)



# ---- implicit-GEMM convolution (csrc/conv.hip) -------------------------------------------------
def _conv_ref(xs, weight, bias=None):
    x = torch.cat([t.float() for t in xs if t is not None], 1)
    return F.conv2d(x, weight.half().float(), bias, padding=weight.shape[-1] // 2)


@pytest.mark.parametrize("n,h,w,ca,cb,nout,k", [(3, 7, 10, 128, 320, 128, 3),     # ragged pixel tile, 2 segments
                                                 (2, 12, 16, 0, 128, 64, 3),       # padded output tile
                                                 (2, 9, 8, 128, 0, 384, 3),        # 3 output tiles
                                                 (5, 6, 6, 64, 64, 576, 1)])       # 1x1
def test_conv_igemm_matches_conv2d(gpu, n, h, w, ca, cb, nout, k):
    from glorie_slam_amd import update_ops as U
    xa = _cl_half(n, ca, h, w, gpu, 11) if ca else None
    wide = _cl_half(n, cb + 64, h, w, gpu, 12) if cb else None
    xb = wide[:, 64:64 + cb] if cb else None                           # a channel slice: row stride != channels
    g = torch.Generator(device="cpu").manual_seed(13)
    weight = (torch.randn(nout, ca + cb, k, k, generator=g) / (3.0 * (ca + cb) ** 0.5)).to(gpu)
    bias = torch.randn(nout, generator=g).to(gpu)
    out = torch.empty((n, nout, h, w), dtype=torch.float16, device=gpu, memory_format=torch.channels_last)
    U.conv_igemm(xa, xb, U.pack_conv_igemm(weight), k * k, nout, out, terms=bias, act=U.ACT_RELU)
    ref = F.relu(_conv_ref([xa, xb], weight, bias))
    torch.testing.assert_close(out.float(), ref, atol=4e-3, rtol=4e-3)
    out2 = torch.empty_like(out)
    U.conv_igemm(xa, xb, U.pack_conv_igemm(weight), k * k, nout, out2)
    torch.testing.assert_close(out2.float(), _conv_ref([xa, xb], weight), atol=4e-3, rtol=4e-3)


def test_conv_igemm_gru_epilogues(gpu):
    from glorie_slam_amd import update_ops as U
    n, h, w = 3, 10, 13
    net = _cl_half(n, 128, h, w, gpu, 21)
    hx = _cl_half(n, 320, h, w, gpu, 22)
    g = torch.Generator(device="cpu").manual_seed(23)
    wzr = (torch.randn(256, 448, 3, 3, generator=g) / 60).to(gpu)
    wq = (torch.randn(128, 448, 3, 3, generator=g) / 60).to(gpu)
    terms = torch.randn(n, 384, generator=g).to(gpu)
    z = torch.empty_like(net)
    rnet = torch.empty_like(net)
    U.conv_igemm(net, hx, U.pack_conv_igemm(wzr), 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, :256],
                 net=net, out2=rnet)
    zr = _conv_ref([net, hx], wzr) + terms[:, :256].view(n, 256, 1, 1)
    zref = torch.sigmoid(zr[:, :128])
    rref = torch.sigmoid(zr[:, 128:]) * net.float()
    torch.testing.assert_close(z.float(), zref, atol=3e-3, rtol=3e-3)
    torch.testing.assert_close(rnet.float(), rref, atol=3e-3, rtol=3e-3)
    new = torch.empty_like(net)
    U.conv_igemm(rnet, hx, U.pack_conv_igemm(wq), 9, 128, new, epilogue=U.EPI_GRU_Q, terms=terms[:, 256:],
                 net=net, z=z)
    q = torch.tanh(_conv_ref([rnet, hx], wq) + terms[:, 256:].reshape(n, 128, 1, 1))
    ref = (1 - z.float()) * net.float() + z.float() * q
    torch.testing.assert_close(new.float(), ref, atol=4e-3, rtol=4e-3)
    # a COPY of packed weights has lost the row-layout flag pack_conv_igemm left on the tensor object: the call must fail
    # loudly (the unpaired epilogue on paired rows would permute the channels silently) unless the caller states the layout
    wq_copy = U.pack_conv_igemm(wq, pair=True).clone()
    with pytest.raises(RuntimeError, match="row-layout flag"):
        U.conv_igemm(rnet, hx, wq_copy, 9, 128, new, epilogue=U.EPI_GRU_Q, terms=terms[:, 256:], net=net, z=z)
    new2 = torch.empty_like(net)
    U.conv_igemm(rnet, hx, wq_copy, 9, 128, new2, epilogue=U.EPI_GRU_Q, terms=terms[:, 256:], net=net, z=z, pair=True)
    assert torch.equal(new2, new)


@pytest.mark.parametrize("mode", ["128", "64", "split", "wide", "nohalo"])
def test_conv_tile_variants_are_bit_identical(gpu, mode):
    """The tile policy only changes which pixels / channels a workgroup owns: every output element sums the same products in
    the same order, so the 64-pixel, split-launch, 128 x 256 and per-tap-staging variants must reproduce the default kernel
    bit for bit - and so must the PAIRED weight packing (a lane owning 8 consecutive channels) against the unpaired one
    (13 maps of 24 x 32 = 9984 pixels: 78 tiles of 128, enough for one whole round + a remainder in the split form)"""
    from glorie_slam_amd import update_ops as U
    n, h, w = 13, 24, 32
    net = _cl_half(n, 128, h, w, gpu, 51)
    wide_t = _cl_half(n, 256, h, w, gpu, 52)
    xb = wide_t[:, 64:256]
    g = torch.Generator(device="cpu").manual_seed(53)
    wq_t = (torch.randn(128, 320, 3, 3, generator=g) / 54).to(gpu)
    wzr_t = (torch.randn(256, 320, 3, 3, generator=g) / 54).to(gpu)
    terms = torch.randn(n, 384, generator=g).to(gpu)
    pre = _cl_half(n, 384, h, w, gpu, 54)
    z0 = _cl_half(n, 128, h, w, gpu, 55).abs().clamp(max=1.0)
    cl = lambda c: torch.empty((n, c, h, w), dtype=torch.float16, device=gpu, memory_format=torch.channels_last)

    def run(policy, pair):
        wq, wzr = U.pack_conv_igemm(wq_t, pair=pair), U.pack_conv_igemm(wzr_t, pair=pair)
        new, z, rnet = cl(128), cl(128), cl(128)
        U.conv_igemm(net, xb, wq, 9, 128, new, epilogue=U.EPI_GRU_Q, terms=terms[:, 256:], net=net, z=z0, pre=pre[:, 256:384],
                     policy=policy)
        U.conv_igemm(net, xb, wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, :256], net=net, out2=rnet, pre=pre[:, 0:256],
                     policy=policy)
        return torch.cat([new, z, rnet], 1).clone()

    ref = run(None, False)
    assert torch.equal(run(mode, False), ref)
    assert torch.equal(run(mode, True), ref) and torch.equal(run(None, True), ref)


# 390 px (ragged second tile) ... G8; on 256 CUs the last three end in a partial round of 64- / 128- / 192-pixel tiles
@pytest.mark.parametrize("n,h,w", [(3, 10, 13), (13, 24, 32), (27, 50, 50), (38, 48, 50), (36, 60, 80)])
def test_conv_pingpong_tiles_are_bit_identical(gpu, n, h, w):
    """conv_pp_kernel (256 channels x 256 pixels, eight waves in two groups half a K-tile apart, DMA pieces as inline asm
    with counted waits) sums the same products in the same order as the default kernel: same bits for the z|r gate launch
    (paired and unpaired weights, with and without the context term) and the plain 320 -> 256 convolution - and the same bits
    again on every repetition (its LDS stages are guarded by barrier counts, not by luck)"""
    from glorie_slam_amd import update_ops as U
    net = _cl_half(n, 128, h, w, gpu, 61)
    wide_t = _cl_half(n, 256, h, w, gpu, 62)
    xb = wide_t[:, 64:256]
    g = torch.Generator(device="cpu").manual_seed(63)
    wzr_t = (torch.randn(256, 320, 3, 3, generator=g) / 54).to(gpu)
    w11_t = (torch.randn(256, 320, 1, 1, generator=g) / 18).to(gpu)
    terms = torch.randn(n, 256, generator=g).to(gpu)
    bias = torch.randn(256, generator=g).to(gpu)
    pre = _cl_half(n, 256, h, w, gpu, 64)
    cl = lambda c: torch.empty((n, c, h, w), dtype=torch.float16, device=gpu, memory_format=torch.channels_last)

    def run(policy, pair, with_pre):
        wzr = U.pack_conv_igemm(wzr_t, pair=pair)
        z, rnet, plain, one = cl(128), cl(128), cl(256), cl(256)
        U.conv_igemm(net, xb, wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms, net=net, out2=rnet,
                     pre=pre if with_pre else None, policy=policy)
        U.conv_igemm(net, xb, wzr, 9, 256, plain, terms=bias, act=U.ACT_RELU, policy=policy)
        U.conv_igemm(net, xb, U.pack_conv_igemm(w11_t, pair=pair), 1, 256, one, terms=bias, policy=policy)
        return torch.cat([z, rnet, plain, one], 1).clone()

    for with_pre in (True, False):
        ref = run("nohalo", False, with_pre)            # conv_igemm_kernel (auto takes the ping-pong tile itself on large maps)
        for rep in range(3):
            assert torch.equal(run("pp", False, with_pre), ref), (with_pre, rep)
        assert torch.equal(run("pp", True, with_pre), ref)
    # layers the tile does not fit are refused, not mangled
    with pytest.raises(Exception):
        U.conv_igemm(net, xb, U.pack_conv_igemm(wzr_t[:128]), 9, 128, cl(128), policy="pp")


# 390 px (one ragged 128-pixel-block tile) ... G8 (256 full 512-pixel tiles + a partial round of 256-pixel tiles on 256 CUs)
# ... and a 56-edge graph (268,800 pixels): from 262,144 pixels on the AUTOMATIC policy takes this kernel for the q gate and the heads
@pytest.mark.parametrize("n,h,w", [(3, 10, 13), (13, 24, 32), (27, 50, 50), (38, 48, 50), (36, 60, 80), (56, 60, 80)])
def test_conv_wide_pingpong_tiles_are_bit_identical(gpu, n, h, w, monkeypatch):
    """conv_ppw_kernel (128 channels x 512 pixels, the ping-pong schedule with each wave group staging its own pixel half)
    against the default kernels: the q gate launch (paired and unpaired weights, with the context term), plain 3x3 layers
    with 128 and 384 output channels over two input segments, a 1x1 layer, and the heads launch (tap GEMMs of two heads in
    the epilogue + the stored GraphAgg third) - same bits, on every repetition"""
    from glorie_slam_amd import update_ops as U
    monkeypatch.delenv("GLORIE_CONV_PPW", raising=False)
    net = _cl_half(n, 128, h, w, gpu, 91)
    wide_t = _cl_half(n, 256, h, w, gpu, 92)
    xb = wide_t[:, 64:256]
    g = torch.Generator(device="cpu").manual_seed(93)
    wq_t = (torch.randn(128, 320, 3, 3, generator=g) / 54).to(gpu)
    w384_t = (torch.randn(384, 128, 3, 3, generator=g) / 34).to(gpu)
    w11_t = (torch.randn(128, 320, 1, 1, generator=g) / 18).to(gpu)
    terms = torch.randn(n, 128, generator=g).to(gpu)
    bias = torch.randn(384, generator=g).to(gpu)
    pre = _cl_half(n, 128, h, w, gpu, 94)
    z0 = _cl_half(n, 128, h, w, gpu, 95).abs().clamp(max=1.0)
    w2 = [(torch.randn(2, 128, 3, 3, generator=g) / 30).to(gpu) for _ in range(2)]
    tapw = U.pack_head_taps(w2)
    cl = lambda c: torch.empty((n, c, h, w), dtype=torch.float16, device=gpu, memory_format=torch.channels_last)

    def run(policy, pair, with_pre):
        wq = U.pack_conv_igemm(wq_t, pair=pair)
        new, plain, wide3, one = cl(128), cl(128), cl(384), cl(128)
        U.conv_igemm(net, xb, wq, 9, 128, new, epilogue=U.EPI_GRU_Q, terms=terms, net=net, z=z0,
                     pre=pre if with_pre else None, policy=policy)
        U.conv_igemm(net, xb, wq, 9, 128, plain, terms=bias[:128].contiguous(), act=U.ACT_RELU, policy=policy)
        U.conv_igemm(net, None, U.pack_conv_igemm(w384_t, pair=pair), 9, 384, wide3, terms=bias, policy=policy)
        U.conv_igemm(net, xb, U.pack_conv_igemm(w11_t, pair=pair), 1, 128, one, terms=bias[:128].contiguous(), policy=policy)
        return torch.cat([new, plain, wide3, one], 1).clone()

    def heads(env):
        if env is None:
            monkeypatch.delenv("GLORIE_CONV_PPW", raising=False)
        else:
            monkeypatch.setenv("GLORIE_CONV_PPW", env)
        rest = cl(128)
        rows = U.conv_igemm_heads(net, U.pack_conv_igemm(w384_t), 9, 384, bias, tapw, 2, out=rest)
        monkeypatch.delenv("GLORIE_CONV_PPW", raising=False)
        return rows.clone(), rest.clone()

    for with_pre in (True, False):
        ref = run("nohalo", False, with_pre)
        for rep in range(3):
            assert torch.equal(run("ppw", False, with_pre), ref), (with_pre, rep)
        assert torch.equal(run("ppw", True, with_pre), ref)
    rows0, rest0 = heads("0")
    for rep in range(3):
        rows1, rest1 = heads("1")
        assert torch.equal(rest1, rest0), rep
        assert torch.equal(rows1, rows0), rep
    rows_a, rest_a = heads(None)                       # the automatic choice (this kernel from 262,144 pixels on)
    assert torch.equal(rest_a, rest0) and torch.equal(rows_a, rows0)
    assert torch.equal(run(None, True, True), run("nohalo", False, True))
    # layers the tile does not fit are refused, not mangled
    with pytest.raises(Exception):
        U.conv_igemm(net, xb, U.pack_conv_igemm(wq_t[:64]), 9, 64, cl(64), policy="ppw")


def test_flow_conv7_matches_conv2d(gpu):
    from glorie_slam_amd import update_ops as U
    n, h, w = 3, 9, 11                                              # 297 pixels: ragged last tile, maps < 7 wide halo
    g = torch.Generator(device="cpu").manual_seed(31)
    flow = (8.0 * torch.randn(n, h, w, 4, generator=g)).to(gpu)
    weight = (torch.randn(128, 4, 7, 7, generator=g) / 14).to(gpu)
    bias = torch.randn(128, generator=g).to(gpu)
    wide = torch.zeros((n, 192, h, w), dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    U.flow_conv7(flow, U.pack_flow_conv7(weight), bias, wide[:, 64:192])
    x = flow.permute(0, 3, 1, 2).half().float()
    ref = F.relu(F.conv2d(x, weight.half().float(), bias, padding=3))
    torch.testing.assert_close(wide[:, 64:192].float(), ref, atol=2e-2, rtol=4e-3)
    assert float(wide[:, :64].abs().max()) == 0.0


@pytest.mark.parametrize("n,h,w", [(3, 9, 11), (1, 7, 7), (5, 12, 16), (36, 60, 80)])
def test_flow_conv7_on_the_padded_fp16_map_is_bit_identical(gpu, n, h, w):
    """flow_encoder[0] on the zero-padded fp16 motion map (one 16-byte load per stencil row is the MFMA fragment) against
    the fp32-map kernel: same fp16 operands, same K order -> same bits; glorie_motion_padded writes what glorie_motion
    + the fp16 rounding give"""
    from glorie_slam_amd import update_ops as U, droid_backends as db
    g = torch.Generator(device="cpu").manual_seed(33)
    flow = (8.0 * torch.randn(n, h, w, 4, generator=g)).to(gpu)
    weight = (torch.randn(128, 4, 7, 7, generator=g) / 14).to(gpu)
    bias = torch.randn(128, generator=g).to(gpu)
    wp = U.pack_flow_conv7(weight)
    a = torch.zeros((n, 128, h, w), dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    wide = torch.zeros((n, 192, h, w), dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    U.flow_conv7(flow, wp, bias, a)
    pf = U.flow_pad(flow, U.PaddedFlow(n, h, w, gpu))
    assert torch.equal(pf.interior(), flow.half())
    assert float(pf.buf.float().abs().sum()) == float(flow.half().float().abs().sum())       # borders stay zero
    U.flow_conv7_padded(pf, wp, bias, wide[:, 64:192])
    assert torch.equal(wide[:, 64:192], a)
    assert float(wide[:, :64].abs().max()) == 0.0
    coords0 = torch.randn(h, w, 2, generator=g).to(gpu) * 5
    coords1 = (coords0 + 40.0 * torch.randn(1, n, h, w, 2, generator=g).to(gpu)).contiguous()
    target = (coords1 + 40.0 * torch.randn(1, n, h, w, 2, generator=g).to(gpu)).contiguous()
    pm = db.motion_padded(coords1, coords0, target, U.PaddedFlow(n, h, w, gpu))
    assert torch.equal(pm.interior(), db.motion(coords1, coords0, target).half())


def test_empty_inputs_are_no_ops(gpu):
    """zero edges / zero frames: every new entry point returns without touching its outputs"""
    from glorie_slam_amd import update_ops as U, droid_backends as db
    e = lambda *s: torch.empty(*s, dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    x, out = e(0, 128, 6, 8), e(0, 128, 6, 8)
    wt = torch.randn(128, 128, 3, 3, device=gpu)
    U.conv_igemm(x, None, U.pack_conv_igemm(wt), 9, 128, out)
    U.bias_act(x, None, U.ACT_RELU)
    assert U.conv3x3_small(x, U.pack_conv3x3_small([wt[:2]]), None, 2, [U.ACT_NONE]).shape == (1, 0, 6, 8, 2)
    assert U.segment_mean(x, torch.zeros(0, dtype=torch.int64, device=gpu), 3).abs().sum() == 0
    flow = torch.empty(0, 6, 8, 4, device=gpu)
    U.flow_conv7(flow, U.pack_flow_conv7(torch.randn(128, 4, 7, 7, device=gpu)), torch.zeros(128, device=gpu), out)
    c = torch.empty(0, 2, 6, 8, device=gpu)
    lv = [db.tile_corr_level(torch.empty(0, 6 >> l, 8 >> l, dtype=torch.float16, device=gpu)) for l in range(3)]
    assert db.corr_lookup_pyramid_tiled(lv, c, 6, 8).shape == (0, 147, 6, 8)
    poses = torch.zeros(4, 7, device=gpu)
    poses[:, 6] = 1
    disps = torch.ones(4, 6, 8, device=gpu)
    intr = torch.tensor([4.0, 4.0, 4.0, 3.0], device=gpu)
    m = db.valid_depth_mask(poses, disps, intr, torch.zeros(0, dtype=torch.int64, device=gpu), 0.01, 2)
    assert m.shape == (0, 6, 8)
    mask = torch.zeros(4, 6, 8, dtype=torch.bool, device=gpu)
    eo, any_on = db.dspo_prepare(poses, disps, intr, disps.clone(), 0, 0.01, 2, 0.1,
                                 torch.zeros(0, dtype=torch.int64, device=gpu), torch.zeros(0, dtype=torch.int64, device=gpu),
                                 mask, torch.ones(4, device=gpu), torch.zeros(4, device=gpu))
    assert eo.numel() == 0 and int(any_on) == 0
    assert db.motion(torch.empty(0, 6, 8, 2, device=gpu), torch.zeros(6, 8, 2, device=gpu),
                     torch.empty(0, 6, 8, 2, device=gpu)).shape == (0, 6, 8, 4)


def test_update_operator_full_size_edge_permutation(gpu):
    """BASELINE size (36 edges of 60x80): the per-edge outputs of the update operator (recurrent state, flow
    revision, confidence) do not depend on the position of the edge in the batch - a permuted batch gives the
    permuted result bit for bit - and the per-keyframe outputs (eta, upsampling mask: means over the edges of a
    source frame, summed in edge order) agree to fp16 rounding."""
    from glorie_slam_amd.droid_net import UpdateModule, FusedUpdate
    torch.manual_seed(13)
    mod = UpdateModule().to(gpu).eval()
    n, h, w = 36, 60, 80
    net, inp, corr, flow = _inputs(gpu, n, h, w, seed=21)
    ii = torch.tensor([i // 5 for i in range(n)], device=gpu)           # 8 source frames
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(2)).to(gpu)
    a = FusedUpdate(mod)(net, inp, corr, flow, ii, None)
    b = FusedUpdate(mod)(net[:, perm].contiguous(), inp[:, perm].contiguous(), corr[:, perm].contiguous(),
                         flow[:, perm].contiguous(), ii[perm], None)
    for name, x, y in zip(["net", "delta", "weight"], a[:3], b[:3]):
        assert torch.equal(x[:, perm], y), name
    for name, x, y in zip(["eta", "upmask"], a[3:], b[3:]):
        torch.testing.assert_close(x.float(), y.float(), atol=4e-3, rtol=4e-3, msg=lambda m, n=name: f"{n}: {m}")
    assert all(torch.isfinite(t.float()).all() for t in a)
