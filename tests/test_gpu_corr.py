"""Parity of the HIP correlation lookups with the oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import corr as ocorr

pytestmark = pytest.mark.gpu


def _vol(rng, N, h1, w1, h2, w2, dtype):
    return rng.standard_normal((N, h1, w1, h2, w2)).astype(dtype)


def _coords(rng, N, h1, w1, h2, w2, margin=5.0):
    return np.stack([rng.uniform(-margin, w2 + margin, (N, h1, w1)),
                     rng.uniform(-margin, h2 + margin, (N, h1, w1))], 1).astype(np.float32)


@pytest.mark.parametrize("shape", [(2, 7, 9, 7, 9), (1, 30, 40, 30, 40), (3, 5, 33, 3, 5)])
def test_corr_index_forward_fp16_bit_exact(gpu, shape):
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(0)
    vol = _vol(rng, *shape, np.float16)
    coords = _coords(rng, *shape)
    # include exact-integer coordinates and far out-of-bounds ones
    coords[0, :, 0, 0] = (2.0, 1.0)
    coords[0, :, 0, 1] = (-50.0, 400.0)
    ref = ocorr.corr_index_forward(vol, coords, 3)
    got, = db.corr_index_forward(torch.from_numpy(vol).to(gpu), torch.from_numpy(coords).to(gpu), 3)
    got = got.cpu().numpy()
    assert got.dtype == np.float16 and got.shape == ref.shape
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), \
        f"max abs diff {np.abs(got.astype(np.float32) - ref.astype(np.float32)).max()}"


def test_corr_index_forward_fp32(gpu):
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(1)
    shape = (2, 6, 10, 12, 14)
    vol = _vol(rng, *shape, np.float32)
    coords = _coords(rng, *shape)
    ref = ocorr.corr_index_forward(vol, coords, 3)
    got, = db.corr_index_forward(torch.from_numpy(vol).to(gpu), torch.from_numpy(coords).to(gpu), 3)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)


def test_corr_generic_radius(gpu):
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(2)
    shape = (1, 4, 5, 9, 9)
    vol = _vol(rng, *shape, np.float16)
    coords = _coords(rng, *shape, margin=2.0)
    ref = ocorr.corr_index_forward(vol, coords, 2)
    got, = db.corr_index_forward(torch.from_numpy(vol).to(gpu), torch.from_numpy(coords).to(gpu), 2)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16))


def test_corr_lookup_pyramid_bit_exact(gpu):
    """fused 4-level lookup == per-level reference path + cat (corr.py:43-53)"""
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(3)
    N, h, w = 2, 30, 40
    levels = [rng.standard_normal((N, h, w, h >> l, w >> l)).astype(np.float16) for l in range(4)]
    coords = _coords(rng, N, h, w, h, w)
    ref = ocorr.corr_lookup_pyramid(levels, coords, 3)
    got = db.corr_lookup_pyramid([torch.from_numpy(v).to(gpu) for v in levels],
                                 torch.from_numpy(coords).to(gpu), 3).cpu().numpy()
    assert got.shape == (N, 196, h, w)
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("N,h,w,margin", [(2, 30, 40, 5.0), (3, 24, 32, 12.0), (1, 60, 80, 3.0), (2, 12, 14, 20.0)])
def test_corr_lookup_tiled_bit_exact(gpu, N, h, w, margin):
    """tiled pyramid (64-byte 4x8 blocks) == row-major pyramid == oracle, incl. windows that leave the
    map on every side and level planes whose size is not a multiple of the block (7x10, 3x3, ...)"""
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(11)
    levels = [rng.standard_normal((N, h, w, h >> l, w >> l)).astype(np.float16) for l in range(4)]
    coords = _coords(rng, N, h, w, h, w, margin=margin)
    coords[0, :, 0, :4] = np.array([[-3.0, -2.5, 0.0, w - 0.25], [-3.0, 0.49, h + 2.0, h - 1.0]], np.float32)
    vols = [torch.from_numpy(v).to(gpu) for v in levels]
    ct = torch.from_numpy(coords).to(gpu)
    plain = db.corr_lookup_pyramid(vols, ct, 3)
    tiled = [db.tile_corr_level(v.view(N * h * w, h >> l, w >> l)) for l, v in enumerate(vols)]
    got = db.corr_lookup_pyramid_tiled(tiled, ct, h, w)
    assert torch.equal(got.view(torch.int16), plain.view(torch.int16))
    ref = ocorr.corr_lookup_pyramid(levels, coords, 3)          # every size, incl. the BASELINE 60x80 planes
    assert np.array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16))
    # channels-last form for the fused update operator: the same values at channel l*64 + dy*8 + dx, padding zero
    if (h * w) % 8 == 0:
        cl = db.corr_lookup_tiled_cl(tiled, ct, h, w)
        assert cl.shape == (N, 256, h, w) and cl.is_contiguous(memory_format=torch.channels_last)
        v = cl.view(N, 4, 8, 8, h, w)                              # [n][level][dy][dx][y][x]
        assert float(v[:, :, 7].abs().max()) == 0.0 and float(v[:, :, :, 7].abs().max()) == 0.0
        planar = got.view(N, 4, 7, 7, h, w)                        # [n][level][dx][dy][y][x]
        assert torch.equal(v[:, :, :7, :7].contiguous().view(torch.int16),
                           planar.permute(0, 1, 3, 2, 4, 5).contiguous().view(torch.int16))
        # coordinates in the reprojection's interleaved [N,h,w,2] layout: same bits, no permute + copy in front
        xy = db.corr_lookup_tiled_cl(tiled, ct.permute(0, 2, 3, 1).contiguous(), h, w, interleaved=True)
        assert torch.equal(xy.view(torch.int16), cl.view(torch.int16))


def test_corrblock_tiled_cat_and_index(gpu):
    """CorrBlock in tiled mode: construction, cat, boolean indexing keep the lookup identical to the
    row-major block"""
    from glorie_slam_amd.droid_net import CorrBlock
    g = torch.Generator(device="cpu").manual_seed(5)
    f1 = torch.randn(1, 5, 128, 16, 24, generator=g).to(gpu).half()
    f2 = torch.randn(1, 5, 128, 16, 24, generator=g).to(gpu).half()
    coords = torch.rand(1, 5, 16, 24, 2, generator=g).to(gpu) * torch.tensor([30.0, 20.0], device=gpu) - 3.0
    a = CorrBlock(f1, f2, tiled=False)
    b = CorrBlock(f1, f2)
    assert b.tiled and not a.tiled
    assert torch.equal(a(coords), b(coords))
    b2 = CorrBlock(f1[:, :3], f2[:, :3]).cat(CorrBlock(f1[:, 3:], f2[:, 3:]))
    assert torch.equal(a(coords), b2(coords))
    keep = torch.tensor([True, False, True, True, False], device=gpu)
    assert torch.equal(a[keep](coords[:, keep]), b[keep](coords[:, keep]))


def test_corr_empty_and_noncontiguous(gpu):
    from glorie_slam_amd import droid_backends as db
    vol = torch.zeros(0, 4, 4, 4, 4, dtype=torch.float16, device=gpu)
    coords = torch.zeros(0, 2, 4, 4, device=gpu)
    out, = db.corr_index_forward(vol, coords, 3)
    assert out.shape == (0, 7, 7, 4, 4)
    vol = torch.zeros(2, 4, 4, 4, 8, dtype=torch.float16, device=gpu)[..., ::2]
    with pytest.raises(RuntimeError, match="contiguous"):
        db.corr_index_forward(vol, torch.zeros(2, 2, 4, 4, device=gpu), 3)


def test_corr_full_size_property(gpu):
    """BASELINE size (60x80, 4 levels): a constant volume must return the constant for every
    in-bounds window (weights sum to 1 up to fp16 rounding) and linearity in the volume."""
    from glorie_slam_amd import droid_backends as db
    N, h, w = 2, 60, 80
    g = torch.Generator(device="cpu").manual_seed(0)
    coords = torch.stack([torch.rand(N, h, w, generator=g) * (w - 20) + 10,
                          torch.rand(N, h, w, generator=g) * (h - 20) + 10], 1).to(gpu)
    levels = [torch.full((N, h, w, h >> l, w >> l), 2.0, dtype=torch.float16, device=gpu) for l in range(4)]
    out = db.corr_lookup_pyramid(levels, coords, 3)
    inner = out[:, :49]  # level 0 windows are fully in bounds for these coords
    assert torch.allclose(inner.float(), torch.full_like(inner.float(), 2.0), atol=4e-3)


def test_altcorr_forward(gpu):
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(4)
    B, H, W, C = 2, 9, 13, 128
    f1 = (rng.standard_normal((B, H, W, C)) * 0.25).astype(np.float32)
    f2 = (rng.standard_normal((B, H // 2, W // 2, C)) * 0.25).astype(np.float32)
    coords = np.stack([rng.uniform(-3, W // 2 + 3, (B, 2, H, W)), rng.uniform(-3, H // 2 + 3, (B, 2, H, W))], -1).astype(np.float32)
    ref = ocorr.altcorr_forward(f1, f2, coords, 3)
    got, = db.altcorr_forward(torch.from_numpy(f1).to(gpu), torch.from_numpy(f2).to(gpu),
                              torch.from_numpy(coords).to(gpu), 3)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


def _otf_inputs(rng, F_, H, W, smooth):
    fm = rng.standard_normal((F_, 128, H, W)).astype(np.float16)
    ii = np.array([0, 1, 2, 0], np.int64) % F_
    jj = np.array([1, 0, 0, 2], np.int64) % F_
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    N = len(ii)
    if smooth:
        coords = np.stack([np.stack([x + 1.7 * n + 0.05 * y, y - 0.9 * n + 0.03 * x], 0) for n in range(N)])
    else:
        coords = np.stack([rng.uniform(-6, W + 6, (N, H, W)), rng.uniform(-6, H + 6, (N, H, W))], 1)
    return fm, coords.astype(np.float32), ii, jj


@pytest.mark.parametrize("smooth", [True, False])
@pytest.mark.parametrize("H,W", [(24, 32), (30, 40), (17, 21)])
def test_corr_otf_matches_oracle(gpu, smooth, H, W):
    """volume-free MFMA lookup vs the oracle's virtual fp16 volume.  Values agree up to the
    fp16 rounding of individual dot products (fp32 MFMA summation order), so the bulk is
    bit-identical and the rest within a few fp16 ulps."""
    from glorie_slam_amd.droid_net import OtfCorrBlock
    rng = np.random.default_rng(5)
    fm, coords, ii, jj = _otf_inputs(rng, 3, H, W, smooth)
    ref = ocorr.corr_otf(fm, coords, ii, jj).astype(np.float32)
    blk = OtfCorrBlock(torch.from_numpy(fm).to(gpu)[None])
    c5 = torch.from_numpy(coords).to(gpu).permute(0, 2, 3, 1)[None].contiguous()
    got = blk(c5, torch.from_numpy(ii).to(gpu), torch.from_numpy(jj).to(gpu))[0].float().cpu().numpy()
    assert got.shape == ref.shape == (len(ii), 196, H, W)
    exact = (got == ref).mean()
    assert exact > 0.97, exact
    np.testing.assert_allclose(got, ref, rtol=4e-3, atol=4e-3)


def test_corr_otf_equals_volume_path(gpu):
    """OTF == CorrBlock volume lookup (reference frontend path) within fp16 tolerance"""
    from glorie_slam_amd.droid_net import OtfCorrBlock, CorrBlock
    rng = np.random.default_rng(6)
    H, W = 24, 32
    fm, coords, ii, jj = _otf_inputs(rng, 3, H, W, True)
    fmt = torch.from_numpy(fm).to(gpu)
    c5 = torch.from_numpy(coords).to(gpu).permute(0, 2, 3, 1)[None].contiguous()
    with torch.autocast("cuda", enabled=True):
        vol = CorrBlock(fmt[torch.from_numpy(ii)][None], fmt[torch.from_numpy(jj)][None])
    a = vol(c5)[0].float()
    b = OtfCorrBlock(fmt[None])(c5, torch.from_numpy(ii).to(gpu), torch.from_numpy(jj).to(gpu))[0].float()
    assert torch.allclose(a, b, rtol=2e-2, atol=1e-2)   # fp16 path rel 2e-2 / abs 1e-2


@pytest.mark.parametrize("smooth,H,W", [(True, 24, 32), (False, 17, 21), (True, 60, 80)])
def test_corr_otf_fused_encoder(gpu, smooth, H, W):
    """glorie_corr_otf_encode: the looked-up features of the same launch (bit-identical to the plain lookup) and
    relu(conv1x1(corr) + b) of corr_encoder[0] (droid_net.py:73-74) against torch on those features"""
    from glorie_slam_amd.droid_net import OtfCorrBlock
    rng = np.random.default_rng(8)
    fm, coords, ii, jj = _otf_inputs(rng, 3, H, W, smooth)
    blk = OtfCorrBlock(torch.from_numpy(fm).to(gpu)[None])
    c5 = torch.from_numpy(coords).to(gpu).permute(0, 2, 3, 1)[None].contiguous()
    it, jt = torch.from_numpy(ii).to(gpu), torch.from_numpy(jj).to(gpu)
    g = torch.Generator().manual_seed(1)
    wgt = (torch.randn(128, 196, 1, 1, generator=g) / 14).to(gpu)
    bias = torch.randn(128, generator=g).to(gpu)
    N = len(ii)
    wide = torch.zeros(N, 320, H, W, dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    plain = blk(c5, it, jt)
    corr = blk.lookup_encode(c5, it, jt, OtfCorrBlock.pack_encoder(wgt), bias, wide[:, 128:256], want_corr=True)
    assert torch.equal(corr, plain)
    ref = torch.relu(torch.nn.functional.conv2d(plain[0].float(), wgt.half().float(), bias))
    torch.testing.assert_close(wide[:, 128:256].float(), ref, rtol=4e-3, atol=4e-3)
    assert float(wide[:, :128].abs().max()) == 0.0 and float(wide[:, 256:].abs().max()) == 0.0
    only = torch.zeros(N, 128, H, W, dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    assert blk.lookup_encode(c5, it, jt, OtfCorrBlock.pack_encoder(wgt), bias, only) is None
    assert torch.equal(only, wide[:, 128:256])


def _fp16_ulp_diff(a, b):
    """0 where two fp16 tensors agree to one fp16 ulp of the reference b (or to 2e-6 absolute: the fp32 summation
    order of a 128-term dot product with partial sums of O(1), visible only where the products cancel to ~0), else
    the excess in ulps"""
    af, bf = a.float(), b.float()
    ulp = torch.exp2(torch.floor(torch.log2(bf.abs().clamp_min(6.2e-5))) - 10.0)
    diff = (af - bf).abs()
    ok = diff <= torch.maximum(ulp, torch.full_like(ulp, 2e-6))
    return torch.where(ok, torch.zeros_like(diff), diff / ulp).to(torch.int32) + (~ok).to(torch.int32)


@pytest.mark.parametrize("layout", ["dm", "tiled"])
@pytest.mark.parametrize("h,w", [(30, 40), (60, 80), (12, 16), (21, 24)])
def test_corr_arena_build_matches_oracle_pyramid(gpu, h, w, layout):
    """glorie_corr_build: level 0 = fp16 all-pairs correlation (one fp16 ulp of the oracle's: fp32 summation order),
    levels 1..3 = exactly avg_pool2d of the level below on the fp16 values; a removed edge frees its slot and the next
    edge is built into it without touching the others; growth keeps every volume"""
    from glorie_slam_amd.droid_net import CorrArena
    from oracle import update_step as ostep
    rng = np.random.default_rng(4)
    F_ = 5
    fm = rng.standard_normal((F_, 128, h, w)).astype(np.float16)
    fcl = (torch.from_numpy(fm).to(gpu) / 4.0).permute(0, 2, 3, 1).reshape(F_, h * w, 128).contiguous()
    ii = np.array([0, 1, 2, 4, 3], np.int64)
    jj = np.array([1, 0, 4, 2, 3], np.int64)
    arena = CorrArena(h, w, gpu, capacity=4, layout=layout)        # 5 edges: grows once
    assert arena.layout == layout
    arena.add(fcl, torch.from_numpy(ii[:3]).to(gpu), torch.from_numpy(jj[:3]).to(gpu))
    arena.add(fcl, torch.from_numpy(ii[3:]).to(gpu), torch.from_numpy(jj[3:]).to(gpu))
    assert len(arena) == 5 and arena.capacity >= 5
    ref = ostep.corr_pyramid_fp16(fm[ii], fm[jj])
    lv = [arena.level(l) for l in range(4)]
    for l in range(4):
        assert tuple(lv[l].shape) == (5, h, w, h >> l, w >> l)
    r0 = torch.from_numpy(ref[0]).to(gpu)
    assert int(_fp16_ulp_diff(lv[0], r0).max()) == 0 and float((lv[0] == r0).float().mean()) > 0.98
    for l in range(1, 4):
        pooled = torch.nn.functional.avg_pool2d(lv[l - 1].reshape(-1, 1, h >> (l - 1), w >> (l - 1)).float(), 2, 2)
        assert torch.equal(lv[l].reshape(pooled.shape).float(), pooled.half().float()), f"level {l}"
        # vs the oracle's pyramid: averages of values that are one ulp apart (absolute, not relative: sums cancel)
        torch.testing.assert_close(lv[l].float(), torch.from_numpy(ref[l]).to(gpu).float(), atol=1e-3, rtol=2e-3)
    # lookups through the slot list == the lookup on the row-major volumes
    from glorie_slam_amd import droid_backends as db
    coords = torch.from_numpy(_coords(rng, 5, h, w, h, w, margin=4.0)).to(gpu)
    got = arena(coords.permute(0, 2, 3, 1)[None].contiguous())[0]
    want = db.corr_lookup_pyramid([v.contiguous() for v in lv], coords, 3)
    assert torch.equal(got, want)
    # remove edges 1 and 3: their slots are recycled by the next additions, survivors keep their data in place
    before = {s_: arena.views()[0].view(arena.capacity, h * w, -1)[s_].clone() for s_ in arena._host_slots}
    freed = [arena._host_slots[1], arena._host_slots[3]]
    arena.keep([True, False, True, False, True])
    assert len(arena) == 3
    arena.add(fcl, torch.tensor([3, 0], device=gpu), torch.tensor([1, 2], device=gpu))
    assert sorted(arena._host_slots[3:]) == sorted(freed)
    for s_ in arena._host_slots[:3]:
        assert torch.equal(arena.views()[0].view(arena.capacity, h * w, -1)[s_], before[s_])
    ref2 = ostep.corr_pyramid_fp16(fm[[3, 0]], fm[[1, 2]])
    assert int(_fp16_ulp_diff(arena.level(0)[3:], torch.from_numpy(ref2[0]).to(gpu)).max()) == 0


def test_corr_arena_against_reference_fixture(gpu):
    """fixture F4 (the reference's CorrBlock pyramid in fp32 on CPU; 16 channels there, so the 128-channel builder is
    fed the fixture's maps zero-extended to 128 channels): fp16 tolerance"""
    import os
    from glorie_slam_amd.droid_net import CorrArena
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corr_pyramid.npz"))
    f1, f2 = f["fmap1"][0], f["fmap2"][0]                  # [2,16,8,8]
    N, C, h, w = f1.shape
    pad = lambda x: np.concatenate([x, np.zeros((N, 128 - C, h, w), x.dtype)], 1)
    maps = np.concatenate([pad(f1), pad(f2)], 0).astype(np.float16)          # frames 0,1 = fmap1; 2,3 = fmap2
    fcl = (torch.from_numpy(maps).to(gpu) / 4.0).permute(0, 2, 3, 1).reshape(2 * N, h * w, 128).contiguous()
    arena = CorrArena(h, w, gpu, num_levels=3)
    arena.add(fcl, torch.tensor([0, 1], device=gpu), torch.tensor([2, 3], device=gpu))
    for l in range(3):
        got = arena.level(l).float().cpu().numpy()
        np.testing.assert_allclose(got, f[f"level{l}"], rtol=4e-3, atol=4e-3)


# ---- displacement-major, source-tiled pyramid (csrc/corr_dm.hip) -----------------------------------------------------
@pytest.mark.parametrize("N,h,w,margin", [(2, 30, 40, 5.0), (3, 24, 32, 12.0), (1, 60, 80, 3.0), (2, 12, 14, 20.0),
                                          (2, 9, 17, 6.0)])
def test_corr_dm_lookup_bit_exact(gpu, N, h, w, margin):
    """the displacement-major lookup == the oracle's lookup on the reference's row-major volumes, bit for bit: windows that
    leave the map on every side, displacements that wrap around the cyclic shift, map sizes that are not multiples of the
    8 x 8 source tile, level planes down to 1 x 2"""
    from glorie_slam_amd import droid_backends as db
    rng = np.random.default_rng(21)
    levels = [rng.standard_normal((N, h, w, h >> l, w >> l)).astype(np.float16) for l in range(4)]
    coords = _coords(rng, N, h, w, h, w, margin=margin)
    coords[0, :, 0, :4] = np.array([[-3.0, -2.5, 0.0, w - 0.25], [-3.0, 0.49, h + 2.0, h - 1.0]], np.float32)
    # smooth flow on the last edge (the regime the layout is built for) incl. exact integers
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    coords[-1] = np.stack([x + 2.0 + 0.05 * y, y - 1.25 + 0.02 * x])
    ref = ocorr.corr_lookup_pyramid(levels, coords, 3)
    vols = [torch.from_numpy(v).to(gpu) for v in levels]
    dm = [db.dm_corr_level(v, l) for l, v in enumerate(vols)]
    for l in range(4):
        assert torch.equal(db.dm_to_rowmajor(dm[l], h, w, l), vols[l])
    ct = torch.from_numpy(coords).to(gpu)
    cl = db.corr_dm_lookup(dm, ct, h, w)
    assert cl.shape == (N, 256, h, w) and cl.is_contiguous(memory_format=torch.channels_last)
    v = cl.view(N, 4, 8, 8, h, w)
    assert float(v[:, :, 7].abs().max()) == 0.0 and float(v[:, :, :, 7].abs().max()) == 0.0
    got = db.cl_to_planar(cl).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), \
        f"{(got.view(np.uint16) != ref.view(np.uint16)).mean()} of the values differ"
    # interleaved coordinates and a slot list (edges stored in another order, with a hole)
    xy = db.corr_dm_lookup(dm, ct.permute(0, 2, 3, 1).contiguous(), h, w, interleaved=True)
    assert torch.equal(xy.view(torch.int16), cl.view(torch.int16))
    perm = torch.randperm(N + 1, generator=torch.Generator().manual_seed(0))[:N]
    store = []
    for l in range(4):
        t = torch.zeros((N + 1, dm[l].shape[1]), dtype=torch.float16, device=gpu)
        t[perm.to(gpu)] = dm[l]
        store.append(t)
    sl = db.corr_dm_lookup(store, ct, h, w, slots=perm.to(gpu).int())
    assert torch.equal(sl.view(torch.int16), cl.view(torch.int16))


@pytest.mark.parametrize("N,h,w", [(3, 24, 32), (2, 21, 19), (2, 60, 80)])
def test_corr_dm_fused_encoder(gpu, N, h, w):
    """corr_encoder[0] as the MFMA epilogue of the lookup launch: relu(conv1x1(corr) + b) (droid_net.py:73-77) against torch
    on the looked-up features; the lookup written by the same launch is bit-identical to the plain one; a channel slice of
    a wider map is written in place"""
    from glorie_slam_amd import droid_backends as db, update_ops as U
    rng = np.random.default_rng(22)
    levels = [torch.from_numpy(rng.standard_normal((N, h, w, h >> l, w >> l)).astype(np.float16)).to(gpu) for l in range(4)]
    dm = [db.dm_corr_level(v, l) for l, v in enumerate(levels)]
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    coords = np.stack([np.stack([x + 1.7 * n + 0.05 * y, y - 0.9 * n + 0.03 * x], -1) for n in range(N)]).astype(np.float32)
    coords[0] += rng.uniform(-4, 4, coords[0].shape).astype(np.float32)
    ct = torch.from_numpy(coords).to(gpu)
    g = torch.Generator().manual_seed(1)
    wgt = (torch.randn(128, 196, 1, 1, generator=g) / 14).to(gpu)
    bias = torch.randn(128, generator=g).to(gpu)
    plain = db.corr_dm_lookup(dm, ct, h, w, interleaved=True)
    wide = torch.zeros(N, 320, h, w, dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    both = db.corr_dm_lookup(dm, ct, h, w, interleaved=True, enc_w=U.pack_corr_encoder_dm(wgt), enc_b=bias,
                             enc_out=wide[:, 128:256])
    assert torch.equal(both.view(torch.int16), plain.view(torch.int16))
    ref = torch.relu(torch.nn.functional.conv2d(db.cl_to_planar(plain).float(), wgt.half().float(), bias))
    torch.testing.assert_close(wide[:, 128:256].float(), ref, rtol=4e-3, atol=4e-3)
    assert float(wide[:, :128].abs().max()) == 0.0 and float(wide[:, 256:].abs().max()) == 0.0
    only = torch.zeros(N, 128, h, w, dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    assert db.corr_dm_lookup(dm, ct, h, w, interleaved=True, want_corr=False, enc_w=U.pack_corr_encoder_dm(wgt),
                             enc_b=bias, enc_out=only) is None
    assert torch.equal(only, wide[:, 128:256])
    # the implicit-GEMM 1x1 launch of rounds 1-2 on the same lookup: same operator, different summation order
    old = U.conv_igemm(plain, None, U.pack_corr_encoder(wgt), 1, 128, torch.empty_like(only), terms=bias, act=U.ACT_RELU)
    torch.testing.assert_close(only.float(), old.float(), rtol=2e-3, atol=2e-3)


def test_corr_dm_lookup_split_into_runs_of_edges(gpu, monkeypatch):
    """the fused launch stores through a buffer descriptor with 32-bit offsets, so glorie_corr_dm_lookup issues calls whose
    output rows span 2 GB or more in runs of edges (coords, slots, lookup and encoder rows advance per run); the run length is
    forced down to 3 edges here (GLORIE_CORR_DM_CHUNK) and everything must equal the single launch bit for bit - with an
    explicit slot list and with the implicit slot = edge"""
    from glorie_slam_amd import droid_backends as db, update_ops as U
    rng = np.random.default_rng(29)
    N, h, w = 8, 16, 24
    levels = [torch.from_numpy(rng.standard_normal((N, h, w, h >> l, w >> l)).astype(np.float16)).to(gpu) for l in range(4)]
    dm = [db.dm_corr_level(v, l) for l, v in enumerate(levels)]
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    coords = np.stack([np.stack([x + 0.7 * n, y - 0.4 * n], -1) for n in range(N)]).astype(np.float32)
    coords += rng.uniform(-2, 2, coords.shape).astype(np.float32)
    ct = torch.from_numpy(coords).to(gpu)
    g = torch.Generator().manual_seed(2)
    ew = U.pack_corr_encoder_dm((torch.randn(128, 196, 1, 1, generator=g) / 14).to(gpu))
    eb = torch.randn(128, generator=g).to(gpu)
    slots = torch.tensor([3, 0, 7, 7, 1, 5, 2, 6], dtype=torch.int32, device=gpu)

    def run(sl):
        out = torch.zeros(N, 128, h, w, dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
        corr = db.corr_dm_lookup(dm, ct, h, w, slots=sl, interleaved=True, enc_w=ew, enc_b=eb, enc_out=out)
        return corr.clone(), out

    for sl in (None, slots):
        monkeypatch.delenv("GLORIE_CORR_DM_CHUNK", raising=False)
        c0, e0 = run(sl)
        monkeypatch.setenv("GLORIE_CORR_DM_CHUNK", "3")
        c1, e1 = run(sl)
        assert torch.equal(c0.view(torch.int16), c1.view(torch.int16)) and torch.equal(e0, e1)
        assert float(e0.abs().max()) > 0


def test_corr_arena_layouts_agree(gpu):
    """the two arena layouts hold the same pyramid: identical lookups (planar and channels-last), and the fused
    lookup + encoder of the displacement-major arena matches the two-launch form of the tiled one"""
    from glorie_slam_amd.droid_net import CorrArena
    from glorie_slam_amd import update_ops as U
    rng = np.random.default_rng(23)
    h, w, F_ = 30, 40, 4
    fm = rng.standard_normal((F_, 128, h, w)).astype(np.float16)
    fcl = (torch.from_numpy(fm).to(gpu) / 4.0).permute(0, 2, 3, 1).reshape(F_, h * w, 128).contiguous()
    ii = torch.tensor([0, 1, 2, 3, 1], device=gpu)
    jj = torch.tensor([1, 0, 3, 2, 2], device=gpu)
    a, b = CorrArena(h, w, gpu, layout="dm"), CorrArena(h, w, gpu, layout="tiled")
    a.add(fcl, ii, jj)
    b.add(fcl, ii, jj)
    for l in range(4):
        assert torch.equal(a.level(l), b.level(l)), f"level {l}"
    coords = torch.from_numpy(_coords(rng, 5, h, w, h, w, margin=4.0)).to(gpu).permute(0, 2, 3, 1)[None].contiguous()
    assert torch.equal(a(coords), b(coords))
    assert torch.equal(a(coords, channels_last=True), b(coords, channels_last=True))
    g = torch.Generator().manual_seed(2)
    wgt = (torch.randn(128, 196, 1, 1, generator=g) / 14).to(gpu)
    bias = torch.randn(128, generator=g).to(gpu)
    out = torch.empty(5, 128, h, w, dtype=torch.float16, device=gpu).contiguous(memory_format=torch.channels_last)
    a.lookup_encode(coords, U.pack_corr_encoder_dm(wgt), bias, out)
    two = U.conv_igemm(b(coords, channels_last=True), None, U.pack_corr_encoder(wgt), 1, 128, torch.empty_like(out),
                       terms=bias, act=U.ACT_RELU)
    torch.testing.assert_close(out.float(), two.float(), rtol=2e-3, atol=2e-3)
