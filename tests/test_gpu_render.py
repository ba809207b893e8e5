"""Parity of the renderer kernels (exact KNN, IDW gather, compositing, decoders) with the
oracle and with the golden fixtures minted from the reference's decoder."""
import os

import numpy as np
import pytest
import torch

from oracle import knn as oknn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _index(gpu, pts, cell=0.08, max_cells=1 << 18):
    from glorie_slam_amd.point_ops import KnnIndex
    idx = KnnIndex(gpu, cell_size=cell, max_cells=max_cells)
    idx.set_points(torch.from_numpy(pts).to(gpu))
    return idx


def _check_knn(gpu, pts, q, cell=0.08, max_cells=1 << 18, radius=0.1):
    idx = _index(gpu, pts, cell, max_cells)
    D, I, nn = idx.search(torch.from_numpy(q).to(gpu), 8, radius=radius)
    rD, rI = oknn.knn_bruteforce(pts, q, 8)
    assert np.array_equal(I.cpu().numpy(), rI), "KNN indices differ from the exact oracle"
    assert np.array_equal(D.cpu().numpy(), rD), "KNN distances differ"
    assert np.array_equal(nn.cpu().numpy(), oknn.neighbor_count(rD, np.float32(radius)))
    return idx


def test_knn_uniform_cloud_bit_exact(gpu):
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    q = rng.uniform(-1.2, 1.2, (3000, 3)).astype(np.float32)
    q[:5] = pts[:5]                      # exact hits (distance 0)
    q[5:10] = (50.0, -30.0, 7.0)         # far outside the grid
    _check_knn(gpu, pts, q)


def test_knn_surface_cloud_and_dynamic_radius(gpu):
    import glorie_slam_amd.synth as synth
    pts, _, _ = synth.box_cloud(n_hits=4000)
    rng = np.random.default_rng(1)
    sel = rng.integers(0, len(pts), 2500)
    q = (pts[sel] * rng.uniform(0.93, 1.07, (2500, 1))).astype(np.float32)
    idx = _check_knn(gpu, pts, q, cell=0.1)
    rad = rng.uniform(0.02, 0.3, 2500).astype(np.float32)
    D, I, nn = idx.search(torch.from_numpy(q).to(gpu), 8, radius_per_query=torch.from_numpy(rad).to(gpu))
    rD, rI = oknn.knn_bruteforce(pts, q, 8)
    assert np.array_equal(nn.cpu().numpy(), oknn.neighbor_count(rD, rad))


def test_ball_bounded_search_counts_like_the_exact_search(gpu):
    """what add_neural_points / sample_near_pcl consume is only the number of neighbours inside the (per-query) radius
    (neural_point.py:165-262, 315-375): the ball-bounded search (weights = (.., ball_only)) must return exactly the counts of
    the unbounded exact search and of brute force - for queries on the surface, off it, and far outside the cloud, where the
    unbounded search walks the grid until it has eight points"""
    import glorie_slam_amd.synth as synth
    pts, _, _ = synth.box_cloud(n_hits=6000)
    rng = np.random.default_rng(4)
    sel = rng.integers(0, len(pts), 4000)
    q = (pts[sel] * rng.uniform(0.9, 1.1, (4000, 1))).astype(np.float32)
    q[:400] = rng.uniform(-6, 6, (400, 3)).astype(np.float32)             # far from every point
    q[400:500] += 0.5
    rad = rng.uniform(0.01, 0.3, 4000).astype(np.float32)
    idx = _index(gpu, pts, 0.1, 1 << 18)
    qt, rt = torch.from_numpy(q).to(gpu), torch.from_numpy(rad).to(gpu)
    nn_exact = idx.search(qt, 8, radius_per_query=rt)[2].cpu().numpy()
    nn_ball = idx.search(qt, 8, radius_per_query=rt, weights=(1, False, True))[2].cpu().numpy()
    rD, _ = oknn.knn_bruteforce(pts, q, 8)
    assert np.array_equal(nn_exact, oknn.neighbor_count(rD, rad))
    assert np.array_equal(nn_ball, nn_exact)
    for r_fixed in (0.04, 0.08):                                          # radius_add / radius_query as scalars
        a = idx.search(qt, 8, radius=r_fixed)[2].cpu().numpy()
        b = idx.search(qt, 8, radius=r_fixed, weights=(1, False, True))[2].cpu().numpy()
        assert np.array_equal(a, b) and np.array_equal(a, oknn.neighbor_count(rD, np.float32(r_fixed)))


def test_knn_ties_duplicates_and_tiny_clouds(gpu):
    rng = np.random.default_rng(2)
    base = rng.uniform(0, 1, (40, 3)).astype(np.float32)
    pts = np.concatenate([base, base, base[:10]])     # duplicated points -> equal distances
    q = rng.uniform(0, 1, (200, 3)).astype(np.float32)
    _check_knn(gpu, pts, q, cell=0.3)
    # integer lattice: many exact ties
    g = np.stack(np.meshgrid(*[np.arange(6)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    qq = (g[::3] + 0.5).astype(np.float32)
    _check_knn(gpu, g, qq, cell=1.0)
    # fewer points than k, and an empty cloud
    idx = _index(gpu, base[:3])
    D, I, nn = idx.search(torch.from_numpy(q[:7]).to(gpu), 8, radius=10.0)
    rD, rI = oknn.knn_bruteforce(base[:3], q[:7], 8)
    assert np.array_equal(I.cpu().numpy(), rI) and np.array_equal(D.cpu().numpy(), rD)
    assert (I.cpu().numpy()[:, 3:] == -1).all() and (nn.cpu().numpy() == 3).all()
    idx = _index(gpu, np.zeros((0, 3), np.float32))
    D, I, nn = idx.search(torch.from_numpy(q[:4]).to(gpu), 8, radius=1.0)
    assert (I.cpu().numpy() == -1).all() and (nn.cpu().numpy() == 0).all()


def test_knn_coarse_grid_forced(gpu):
    """max_cells tiny -> the device enlarges the cell size; results must not change"""
    rng = np.random.default_rng(3)
    pts = rng.normal(0, 1, (3000, 3)).astype(np.float32)
    q = rng.normal(0, 1.5, (500, 3)).astype(np.float32)
    _check_knn(gpu, pts, q, cell=0.01, max_cells=64)


def test_knn_ray_ordered_queries_cooperative_path(gpu):
    """Consecutive queries = consecutive samples of neighbouring rays (the renderer's order): the waves take
    the cooperative box scan.  Also a ragged tail (Q not a multiple of 64) and a mixed batch in which
    some waves are coherent and some are not."""
    import glorie_slam_amd.synth as synth
    pts, _, _ = synth.box_cloud(n_hits=30000)
    ro, rd, depth, _, _ = synth.box_rays(H=48, W=64, fx=32.0, fy=32.0, cx=31.5, cy=23.5)
    sel = np.arange(20 * 64, 20 * 64 + 333)                       # 333 neighbouring pixels
    z = depth[sel, None] * np.linspace(0.95, 1.05, 10, dtype=np.float32)[None]
    q = (ro[sel, None] + rd[sel, None] * z[..., None]).reshape(-1, 3).astype(np.float32)
    _check_knn(gpu, pts, q, cell=0.08)
    _check_knn(gpu, pts, q[:1001], cell=0.05)
    rng = np.random.default_rng(7)
    mixed = np.concatenate([q[:640], rng.uniform(-3, 3, (200, 3)).astype(np.float32), q[640:1500]])
    _check_knn(gpu, pts, mixed, cell=0.08)


@pytest.mark.parametrize("rows,W,S", [(48, 64, 10), (37, 50, 10), (5, 19, 3), (16, 16, 1)])
def test_knn_image_layout_is_only_an_ordering(gpu, rows, W, S):
    """glorie_knn_query_image: the patch-wise walk over an image strip (ragged widths, a partial last row) writes
    the same (D, I, nn) into the same rows as the plain query, and both equal the brute-force oracle."""
    import glorie_slam_amd.synth as synth
    pts, _, _ = synth.box_cloud(n_hits=30000)
    ro, rd, depth, _, _ = synth.box_rays(H=48, W=64, fx=32.0, fy=32.0, cx=31.5, cy=23.5)
    yy, xx = np.meshgrid(np.arange(rows), np.arange(W), indexing="ij")
    sel = (yy * 64 + xx % 64).reshape(-1)
    sel = sel[:len(sel) - 7] if rows == 37 else sel               # partial last row
    z = depth[sel, None] * np.linspace(0.95, 1.05, S, dtype=np.float32)[None]
    q = (ro[sel, None] + rd[sel, None] * z[..., None]).reshape(-1, 3).astype(np.float32)
    idx = _index(gpu, pts)
    qd = torch.from_numpy(q).to(gpu)
    rad = torch.from_numpy(np.random.default_rng(2).uniform(0.03, 0.2, len(q)).astype(np.float32)).to(gpu)
    D0, I0, n0 = idx.search(qd, 8, radius_per_query=rad)
    D1, I1, n1 = idx.search(qd, 8, radius_per_query=rad, image_layout=(S, W))
    assert torch.equal(D0, D1) and torch.equal(I0, I1) and torch.equal(n0, n1)
    if len(q) <= 20000:
        rD, rI = oknn.knn_bruteforce(pts, q, 8)
        assert np.array_equal(I1.cpu().numpy(), rI) and np.array_equal(D1.cpu().numpy(), rD)
    with pytest.raises(RuntimeError):
        idx.search(qd[:(S + 1) * 3 + 1], 8, radius=0.1, image_layout=(S + 1, W))     # Q is not a multiple of S


def test_knn_point_permutation_invariance(gpu):
    rng = np.random.default_rng(4)
    pts = rng.uniform(0, 1, (2000, 3)).astype(np.float32)
    q = rng.uniform(0, 1, (300, 3)).astype(np.float32)
    perm = rng.permutation(len(pts))
    D1, I1, _ = _index(gpu, pts).search(torch.from_numpy(q).to(gpu), 8, radius=0.1)
    D2, I2, _ = _index(gpu, pts[perm]).search(torch.from_numpy(q).to(gpu), 8, radius=0.1)
    assert np.array_equal(D1.cpu().numpy(), D2.cpu().numpy())
    assert np.array_equal(np.sort(I1.cpu().numpy(), 1), np.sort(perm[I2.cpu().numpy()], 1))


def test_idw_gather(gpu):
    from glorie_slam_amd import point_ops
    rng = np.random.default_rng(5)
    pts = rng.uniform(0, 1, (3000, 3)).astype(np.float32)
    feats = rng.normal(0, 0.1, (3000, 32)).astype(np.float32)
    q = rng.uniform(-0.1, 1.1, (1000, 3)).astype(np.float32)
    rad = rng.uniform(0.02, 0.15, 1000).astype(np.float32)
    idx = _index(gpu, pts)
    D, I, nn = idx.search(torch.from_numpy(q).to(gpu), 8, radius_per_query=torch.from_numpy(rad).to(gpu))
    c, has, w = point_ops.idw_gather(D, I, nn, torch.from_numpy(feats).to(gpu),
                                     radius_per_query=torch.from_numpy(rad).to(gpu), return_weights=True)
    rc, rhas, rw = oknn.idw_gather(D.cpu().numpy(), I.cpu().numpy(), nn.cpu().numpy(), feats, rad)
    assert np.array_equal(has.cpu().numpy(), rhas) and 0 < rhas.sum() < len(rhas)
    np.testing.assert_allclose(w.cpu().numpy(), rw, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(c.cpu().numpy(), rc, rtol=1e-4, atol=1e-6)
    # both tables in one pass: the bits of two separate calls
    feats_b = rng.normal(0, 0.1, (3000, 32)).astype(np.float32)
    cb, _ = point_ops.idw_gather(D, I, nn, torch.from_numpy(feats_b).to(gpu), radius_per_query=torch.from_numpy(rad).to(gpu))
    c2a, c2b, has2, w2 = point_ops.idw_gather2(D, I, nn, torch.from_numpy(feats).to(gpu), torch.from_numpy(feats_b).to(gpu),
                                               radius_per_query=torch.from_numpy(rad).to(gpu))
    assert torch.equal(c2a, c) and torch.equal(c2b, cb) and torch.equal(has2, has) and torch.equal(w2, w)


@pytest.mark.parametrize("image", [False, True])
def test_knn_query_with_weights_equals_the_separate_weights_launch(gpu, image):
    """glorie_knn_query_weights: the IDW weights and the neighbour mask written by the search launch are the bits of
    glorie_idw_gather on the search's own output (fixed and per-query radius, both query orders, min_nn 1..3), and the
    search results themselves do not change"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd import point_ops
    pts, _, _ = synth.box_cloud(n_hits=30000)
    ro, rd, depth, _, _ = synth.box_rays(H=48, W=64, fx=32.0, fy=32.0, cx=31.5, cy=23.5)
    S, W = 10, 64
    z = depth[:, None] * np.linspace(0.95, 1.05, S, dtype=np.float32)[None]
    q = (ro[:, None] + rd[:, None] * z[..., None]).reshape(-1, 3).astype(np.float32)
    q[::37] += 5.0                                                        # samples without any neighbour in range
    idx = _index(gpu, pts)
    qd = torch.from_numpy(q).to(gpu)
    rad = torch.from_numpy(np.random.default_rng(3).uniform(0.02, 0.2, len(q)).astype(np.float32)).to(gpu)
    layout = (S, W) if image else None
    for kw in (dict(radius_per_query=rad), dict(radius=0.07)):
        D0, I0, n0 = idx.search(qd, 8, image_layout=layout, **kw)
        for min_nn in (1, 2, 3):
            D1, I1, n1, w1, has1 = idx.search(qd, 8, image_layout=layout, weights=(min_nn, False), **kw)
            assert torch.equal(D0, D1) and torch.equal(I0, I1) and torch.equal(n0, n1)
            _, has0, w0 = point_ops.idw_gather(D0, I0, n0, None, radius=kw.get("radius", 0.0),
                                               radius_per_query=kw.get("radius_per_query"), min_nn=min_nn,
                                               return_weights=True, raw_mask=True)
            assert torch.equal(w1, w0) and torch.equal(has1, has0)
            assert 0 < int(has1.sum()) < len(q)
            # the search bounded by the query radius (what the renderer asks for): weights, mask, counts and every slot
            # inside the ball are those of the exact search; only slots beyond the radius (weight 0) may differ
            D2, I2, n2, w2, has2 = idx.search(qd, 8, image_layout=layout, weights=(min_nn, False, True), **kw)
            assert torch.equal(w2, w0) and torch.equal(has2, has0) and torch.equal(n2, n0)
            r2 = (kw["radius_per_query"] if "radius_per_query" in kw else torch.full_like(rad, kw["radius"])) ** 2
            inside = D0 <= r2[:, None]
            assert torch.equal(D2[inside], D0[inside]) and torch.equal(I2[inside], I0[inside])
            assert bool((~inside).any()) and bool(((D2 > r2[:, None]) | (I2 < 0))[~inside].all())


def test_composite_matches_reference_fixture(gpu):
    from glorie_slam_amd import point_ops
    f = np.load(os.path.join(GOLD, "raw2outputs.npz"))
    d, v, c, w = point_ops.composite(torch.from_numpy(f["raw"]).to(gpu), torch.from_numpy(f["z"]).to(gpu), 0.1)
    np.testing.assert_allclose(d.cpu().numpy(), f["depth"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v.cpu().numpy(), f["var"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(c.cpu().numpy(), f["rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(w.cpu().numpy(), f["weights"], rtol=1e-5, atol=1e-7)


def _cfg(dev):
    return {"device": str(dev),
            "pointcloud": {"nn_weighting": "distance", "use_dynamic_radius": True, "min_nn_num": 2,
                           "nn_num": 8, "radius_query": 0.08, "radius_add": 0.04, "radius_min": 0.02},
            "rendering": {"N_surface": 10, "near_end_surface": 0.95, "far_end_surface": 1.05,
                          "sample_near_pcl": True, "sigmoid_coef": 0.1, "near_end": 0.3},
            "model": {"encode_rel_pos_in_col": True, "encode_viewd": True, "c_dim": 32}}


def test_decoders_on_gpu_match_reference_fixture(gpu):
    """POINT.forward through the HIP KNN + IDW kernels == the reference decoder outputs"""
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    f = np.load(os.path.join(GOLD, "decoders.npz"))
    t = lambda k: torch.from_numpy(f[k]).to(gpu)
    torch.manual_seed(43)
    dec = POINT(_cfg(gpu), c_dim=32, hidden_size=128, use_view_direction=True).eval().to(gpu)
    npc = NeuralPointCloud(_cfg(gpu))
    npc.add_points(t("cloud"), t("geo"), t("col"))
    with torch.no_grad():
        raw, ray_mask, point_mask, counter = dec(t("p")[None], npc, "color", npc.geo_feats, npc.col_feats,
                                                 pts_num=10, cloud_pos=npc.cloud_pos(),
                                                 pts_views_d=t("views"), dynamic_r_query=t("radius"))
    pm = f["point_mask"]
    assert np.array_equal(point_mask.cpu().numpy(), pm)
    assert np.array_equal(ray_mask.cpu().numpy(), f["ray_mask"])
    assert np.array_equal(counter.cpu().numpy(), f["counter"])
    raw = raw.cpu().numpy()
    np.testing.assert_allclose(raw[pm, 3], f["occ"][pm], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(raw[pm, :3], f["rgb"][pm], rtol=1e-3, atol=1e-3)   # colours: 1e-3


def test_decoders_on_the_fp16_matrix_cores_keep_fp32_accuracy(gpu, monkeypatch):
    """the three decoders as 3-term hi/lo splits on the fp16 MFMA (mlp_geo_v4 / mlp_nb_v4 / mlp_col_v4, the default) against
    the fp32-MFMA kernels (GLORIE_MLP_F32=1): 2e-6 on colours in [0, 1], 1e-5 on occupancy logits - two orders below the
    fixture tolerance"""
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    f = np.load(os.path.join(GOLD, "decoders.npz"))
    t = lambda k: torch.from_numpy(f[k]).to(gpu)
    torch.manual_seed(43)
    dec = POINT(_cfg(gpu), c_dim=32, hidden_size=128, use_view_direction=True).eval().to(gpu)
    npc = NeuralPointCloud(_cfg(gpu))
    npc.add_points(t("cloud"), t("geo"), t("col"))
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GLORIE_MLP_F32", mode)
        with torch.no_grad():
            raw, _, point_mask, _ = dec(t("p")[None], npc, "color", npc.geo_feats, npc.col_feats, pts_num=10,
                                        cloud_pos=npc.cloud_pos(), pts_views_d=t("views"), dynamic_r_query=t("radius"))
        outs.append(raw.cpu().numpy())
    pm = f["point_mask"]
    np.testing.assert_allclose(outs[1][pm, 3], outs[0][pm, 3], rtol=0, atol=1e-5)      # occupancy logits in [-3, 2]
    np.testing.assert_allclose(outs[1][pm, :3], outs[0][pm, :3], rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[1][pm, :3], f["rgb"][pm], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("what", ["weights", "features", "positions"])
def test_split_decoders_over_six_decades_of_magnitude(gpu, what):
    """the 3-term fp16 split against the exact-fp32 MFMA kernels with the decoder weights, the point features or the scene
    scaled by 1e-3 ... 1e3 (the reference loads a pretrained middle_fine.pt and trains the features: magnitudes are not those of
    the default init).  While every operand fits fp16 the two agree to 1e-5 on colours (1e-4 above the default magnitudes)
    and 2e-5 * max(1, max |occ|) on the occupancy logits (below 2^-3 the low half of the split is an fp16 subnormal: the error is absolute there, 2^-25 per
    operand); when an operand leaves the fp16 range (|x| > 65504) the range guard trips and POINT.forward / the renderer
    answer with the fp32 kernels' values - never with an inf, a NaN or silent garbage."""
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd import point_ops
    f = np.load(os.path.join(GOLD, "decoders.npz"))
    t = lambda k: torch.from_numpy(f[k]).to(gpu)
    pm = f["point_mask"]
    tripped_any = False
    # (features of 1e6 x N(0, 0.1) are beyond fp16 themselves: the guard must trip there at the latest)
    for scale in (1e-3, 1e-2, 1e-1, 1.0, 1e1, 1e2, 1e3) + ((1e6,) if what == "features" else ()):
        torch.manual_seed(43)
        dec = POINT(_cfg(gpu), c_dim=32, hidden_size=128, use_view_direction=True).eval().to(gpu)
        geo, col, cloud, p, radius = t("geo"), t("col"), t("cloud"), t("p"), t("radius")
        if what == "weights":
            with torch.no_grad():
                for name, prm in dec.named_parameters():
                    if name.endswith("weight"):
                        prm.mul_(scale)
        elif what == "features":
            geo, col = geo * scale, col * scale
        else:
            cloud, p, radius = cloud * scale, p * scale, radius * scale
        npc = NeuralPointCloud(_cfg(gpu))
        npc.add_points(cloud, geo, col)
        args = (p[None], npc, "color", npc.geo_feats, npc.col_feats)
        kw = dict(pts_num=10, cloud_pos=npc.cloud_pos(), pts_views_d=t("views"), dynamic_r_query=radius)
        # what the kernels do on their own: the split with its guard, and the exact kernels
        D, I, nn = npc.find_neighbors_faiss(p.reshape(-1, 3).clone(), step="query", dynamic_radius=radius)
        c_geo, has, w = point_ops.idw_gather(D, I, nn, npc.geo_feats, radius=0.0, radius_per_query=radius, min_nn=2,
                                             return_weights=True)
        guard = dec.range_guard(gpu)
        call = lambda **k: point_ops.render_mlp(dec._packed(), p.reshape(-1, 3), t("views"), npc.cloud_pos(), npc.col_feats,
                                                c_geo, I, w, has, stage="color", **k)
        split = call(range_flag=guard.flag).cpu().numpy()
        tripped = guard.tripped()
        exact = call(precise=True).cpu().numpy()
        assert np.isfinite(exact).all()
        hm = has.cpu().numpy().astype(bool)
        if not tripped:
            assert np.isfinite(split).all(), (what, scale)
            # 22 mantissa bits relative to the largest term of a dot product: the scale of a network's values, not of one output
            big = max(1.0, float(np.abs(exact[hm, 3]).max()))
            np.testing.assert_allclose(split[hm, :3], exact[hm, :3], rtol=0, atol=1e-5 if scale <= 1.0 else 2e-4,
                                       err_msg=f"{what} x {scale}")
            assert float(np.abs(split[hm, 3] - exact[hm, 3]).max()) <= 2e-5 * big, (what, scale)
        tripped_any |= tripped
        # what the product path returns: always the exact kernels' values where the guard trips
        with torch.no_grad():
            raw, _, point_mask, _ = dec(*args, **kw)
        raw = raw.cpu().numpy()
        assert np.isfinite(raw).all(), (what, scale)
        if tripped:
            np.testing.assert_array_equal(raw[hm], exact[hm])
        assert not guard.tripped()                                   # POINT.forward consumed its own trip
    if what in ("weights", "features"):
        assert tripped_any, "1e3 x the default magnitudes must leave the fp16 range somewhere"


def test_render_batch_ray_end_to_end(gpu):
    """render a small view of the synthetic box; rays that hit the cloud are valid, depth is
    close to the surface depth, zero-depth rays go through sample_near_pcl"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd.renderer import Renderer
    cfg = _cfg(gpu)
    pts, geo, col = synth.box_cloud(n_hits=40000)
    H, W = 24, 32
    ro, rd, depth, radius, c2w = synth.box_rays(H, W, fx=16.0, fy=16.0, cx=15.5, cy=11.5)
    t = lambda x: torch.from_numpy(x).to(gpu)
    npc = NeuralPointCloud(cfg)
    npc.add_points(t(pts), t(geo), t(col))
    torch.manual_seed(43)
    dec = POINT(cfg, use_view_direction=True).eval().to(gpu)

    class Cam:
        pass
    cam = Cam()
    cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy = H, W, 16.0, 16.0, 15.5, 11.5
    ren = Renderer(cfg, cam)
    gt = depth.copy()
    gt[::9] = 0.0
    with torch.no_grad():
        d, u, c, vm, cnt = ren.render_batch_ray(npc, dec, t(rd), t(ro), gpu, "color", gt_depth=t(gt),
                                                npc_geo_feats=npc.geo_feats, npc_col_feats=npc.col_feats,
                                                cloud_pos=npc.cloud_pos(), dynamic_r_query=t(radius * 2.5))
    assert d.shape == (H * W,) and c.shape == (H * W, 3) and vm.dtype == torch.bool
    assert vm.float().mean() > 0.9
    ok = vm.cpu().numpy() & (gt > 0)
    assert np.all(np.isfinite(d.cpu().numpy()))
    assert np.abs(d.cpu().numpy()[ok] - depth[ok]).max() < 0.06 * depth.max()
    assert (c.cpu().numpy() >= 0).all() and (c.cpu().numpy() <= 1).all()


def test_render_batch_ray_deferred_range_guard(gpu):
    """defer_guard=True: same values, no guard read behind the batch - a trip stays raised for the caller, who renders again
    without the flag and gets the exact-fp32 kernels' values (what bench.render_pass and Renderer.render_img do per frame)"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd.renderer import Renderer
    cfg = _cfg(gpu)
    pts, geo, col = synth.box_cloud(n_hits=40000)
    H, W = 24, 32
    ro, rd, depth, radius, c2w = synth.box_rays(H, W, fx=16.0, fy=16.0, cx=15.5, cy=11.5)
    t = lambda x: torch.from_numpy(x).to(gpu)

    class Cam:
        pass
    cam = Cam()
    cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy = H, W, 16.0, 16.0, 15.5, 11.5
    ren = Renderer(cfg, cam)
    for feat_scale, expect_trip in ((1.0, False), (1e6, True)):
        npc = NeuralPointCloud(cfg)
        npc.add_points(t(pts), t(geo) * feat_scale, t(col) * feat_scale)
        torch.manual_seed(43)
        dec = POINT(cfg, use_view_direction=True).eval().to(gpu)
        run = lambda **k: ren.render_batch_ray(npc, dec, t(rd), t(ro), gpu, "color", gt_depth=t(depth),
                                               npc_geo_feats=npc.geo_feats, npc_col_feats=npc.col_feats,
                                               cloud_pos=npc.cloud_pos(), dynamic_r_query=t(radius * 2.5), **k)
        with torch.no_grad():
            checked = run()
            assert not dec.range_guard(gpu).tripped()                # consumed by the per-batch check
            deferred = run(defer_guard=True)
            assert dec.range_guard(gpu).tripped() == expect_trip     # left for the caller
            if not expect_trip:
                for a, b in zip(checked, deferred):
                    assert torch.equal(a, b)
            else:
                again = run()                                        # the caller's answer to a trip
                for a, b in zip(checked, again):
                    assert torch.equal(a, b)
                assert all(bool(torch.isfinite(a.float()).all()) for a in again)


def _reference_render_setup(gpu, ray_batch_size=65536):
    """scene, decoders and renderer of fixture F11 (tests/golden/render.npz, minted from the reference's
    Renderer by tests/golden/make_golden.py::make_render)"""
    from golden.scenes import render_scene, render_cfg
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd.renderer import Renderer
    f = np.load(os.path.join(GOLD, "render.npz"))
    cloud, geo, col, c2w, cam, depth, depth_zero, radius = render_scene()
    sums = [cloud.double().sum(), geo.double().abs().sum(), col.double().abs().sum(), depth.double().sum(),
            radius.double().sum()]
    np.testing.assert_allclose([float(v) for v in sums], f["scene_sums"], rtol=1e-12)
    cfg = dict(_cfg(gpu))
    cfg["rendering"] = render_cfg()["rendering"]
    torch.manual_seed(43)
    dec = POINT(cfg, c_dim=32, hidden_size=128, use_view_direction=True).eval()
    psum = float(sum(p.detach().double().abs().sum() for p in dec.parameters()))
    assert abs(psum - float(f["param_abs_sum"])) < 1e-9 * psum           # the fixture's decoder weights
    dec = dec.to(gpu)
    npc = NeuralPointCloud(cfg)
    npc.add_points(cloud.to(gpu), geo.to(gpu), col.to(gpu))

    class Cam:
        pass
    c = Cam()
    for k, v in cam.items():
        setattr(c, k, v)
    ren = Renderer(cfg, c, ray_batch_size=ray_batch_size)
    t = lambda k: torch.from_numpy(f[k]).to(gpu)
    return f, npc, dec, ren, t("rays_o"), t("rays_d"), c2w.to(gpu), depth.to(gpu), depth_zero.to(gpu), radius.to(gpu)


def _check_render(got, f, tag, sel=None):
    """rendered colours 1e-3; depth / variance rel 1e-3; masks and counts exact"""
    d, u, c, vm, cnt = [x.detach().cpu().numpy() for x in got]
    pick = (lambda a: a) if sel is None else (lambda a: a[sel])
    ref = {k: pick(f[f"{tag}_{k}"].reshape((-1, 3) if k == "color" else (-1,)))
           for k in ("depth", "unc", "color", "mask", "count")}
    assert np.array_equal(pick(vm.reshape(-1)).astype(bool), ref["mask"].astype(bool)), tag + " valid_ray_mask"
    assert np.array_equal(pick(cnt.reshape(-1)).astype(np.int64), ref["count"].astype(np.int64)), tag + " counts"
    # a ray none of whose samples has neighbours is coloured from the reference's RANDOM placeholder features
    # (decoder.py:170-171,386-387: N(0, 0.01) per call) - not reproducible, and masked invalid either way
    seen = ref["count"] > 0
    np.testing.assert_allclose(pick(c.reshape(-1, 3))[seen], ref["color"][seen], atol=1e-3, err_msg=tag + " colour")
    np.testing.assert_allclose(pick(d.reshape(-1)), ref["depth"], rtol=1e-3, atol=1e-4, err_msg=tag + " depth")
    np.testing.assert_allclose(pick(u.reshape(-1)), ref["unc"], rtol=5e-3, atol=1e-5, err_msg=tag + " uncertainty")


def test_render_batch_ray_matches_reference_renderer(gpu):
    """Renderer.render_batch_ray (Renderer.py:80-219) against the outputs of the REFERENCE's Renderer on the
    same cloud / decoders / rays (fixture F11): the HIP-only fast path (every ray has a depth prior), the
    general path, and the batch with zero-depth rays (sample_near_pcl branch, neural_point.py:315-375)."""
    f, npc, dec, ren, ro, rd, c2w, depth, depth_zero, radius = _reference_render_setup(gpu)
    kw = dict(npc_geo_feats=npc.geo_feats, npc_col_feats=npc.col_feats, cloud_pos=npc.cloud_pos(),
              dynamic_r_query=radius)
    with torch.no_grad():
        ren.use_fast_path = True
        _check_render(ren.render_batch_ray(npc, dec, rd, ro, gpu, "color", gt_depth=depth, **kw), f, "a")
        ren.use_fast_path = False
        _check_render(ren.render_batch_ray(npc, dec, rd, ro, gpu, "color", gt_depth=depth, **kw), f, "a")
        ren.use_fast_path = True
        _check_render(ren.render_batch_ray(npc, dec, rd, ro, gpu, "color", gt_depth=depth_zero, **kw), f, "b")
    assert 100 < int(f["a_mask"].sum()) < 192 and int((f["a_count"] == 0).sum()) > 10   # a non-trivial fixture


def test_render_img_matches_reference_renderer(gpu):
    """Renderer.render_img (Renderer.py:222-306) with the reference's 50-ray batches: get_rays, per-batch `far`,
    float64 outputs -- against the image the reference rendered (fixture F11)"""
    f, npc, dec, ren, ro, rd, c2w, depth, depth_zero, radius = _reference_render_setup(gpu, ray_batch_size=50)
    H, W = ren.H, ren.W
    from glorie_slam_amd.common import get_rays
    go, gd = get_rays(H, W, ren.fx, ren.fy, ren.cx, ren.cy, c2w, gpu)         # R7 on the device == reference rays
    np.testing.assert_allclose(gd.reshape(-1, 3).cpu().numpy(), f["rays_d"], atol=1e-6)
    np.testing.assert_allclose(go.reshape(-1, 3).cpu().numpy(), f["rays_o"], atol=1e-6)
    out = ren.render_img(npc, dec, c2w, gpu, "color", gt_depth=depth_zero.reshape(H, W), npc_geo_feats=npc.geo_feats,
                         npc_col_feats=npc.col_feats, dynamic_r_query=radius.reshape(H, W), cloud_pos=npc.cloud_pos())
    assert out[0].dtype == torch.float64 and out[0].shape == (H, W) and out[2].shape == (H, W, 3)
    _check_render(out, f, "img")


def test_render_img_strips_on_two_streams_equal_one_batch(gpu):
    """Renderer.render_img with strips of 16 image rows on the renderer's two batch streams (four strips in flight two at a
    time) against the same frame rendered as one batch: bit-identical, the zero-depth fallback of a strip included"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd.renderer import Renderer
    cfg = _cfg(gpu)
    pts, geo, col = synth.box_cloud(n_hits=40000)
    H, W = 64, 32
    ro, rd, depth, radius, c2w = synth.box_rays(H, W, fx=16.0, fy=16.0, cx=15.5, cy=31.5)
    t = lambda x: torch.from_numpy(x).to(gpu)
    npc = NeuralPointCloud(cfg)
    npc.add_points(t(pts), t(geo), t(col))
    torch.manual_seed(43)
    dec = POINT(cfg, use_view_direction=True).eval().to(gpu)

    class Cam:
        pass
    cam = Cam()
    cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy = H, W, 16.0, 16.0, 15.5, 31.5
    strips, whole = Renderer(cfg, cam, ray_batch_size=16 * W), Renderer(cfg, cam, ray_batch_size=H * W)
    for zero_at in (None, 16 * W * 2 + 7):
        gt = t(depth).clone()
        if zero_at is not None:
            gt[zero_at] = 0.0                                   # the third strip takes the general path
        kw = dict(gt_depth=gt.reshape(H, W), npc_geo_feats=npc.geo_feats, npc_col_feats=npc.col_feats,
                  dynamic_r_query=t(radius * 2.5).reshape(H, W), cloud_pos=npc.cloud_pos())
        with torch.no_grad():
            a = strips.render_img(npc, dec, t(c2w), gpu, "color", **kw)
            b = whole.render_img(npc, dec, t(c2w), gpu, "color", **kw)
        if zero_at is None:
            assert strips._streams is not None                  # the strips did go through the batch streams
            for x, y in zip(a, b):
                assert torch.equal(x, y)
        else:
            # (a whole frame with one zero-depth ray takes the general path everywhere: torch arithmetic, 1e-5)
            for x, y in zip(a, b):
                torch.testing.assert_close(x.float(), y.float(), atol=2e-4, rtol=1e-4)


def test_ray_samples_camera_equals_get_rays_plus_ray_samples(gpu):
    """get_rays fused into the sample placement (row R7 + R4): the rays of a pixel strip formed in the kernel give the bits of
    get_rays followed by ray_samples, for a rotated camera and a strip that starts and ends inside image rows"""
    from glorie_slam_amd import point_ops
    from glorie_slam_amd.common import get_rays
    H, W, fx, fy, cx, cy = 48, 64, 51.3, 49.7, 31.2, 23.9
    ang = np.array([0.3, -0.7, 1.1])
    from scipy.spatial.transform import Rotation as Rot
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = Rot.from_rotvec(ang).as_matrix().astype(np.float32)
    c2w[:3, 3] = [0.4, -1.3, 2.2]
    ro, rd = get_rays(H, W, fx, fy, cx, cy, torch.from_numpy(c2w), gpu)
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    g = torch.Generator(device="cpu").manual_seed(3)
    first, R = 64 * 5 + 17, 64 * 20 + 9
    depth = (torch.rand(R, generator=g) * 3 + 0.5).to(gpu)
    depth[5] = 0.0
    radius = (torch.rand(R, generator=g) * 0.1 + 0.02).to(gpu)
    a = point_ops.ray_samples(ro[first:first + R], rd[first:first + R], depth, radius, 10, 0.95, 1.05)
    cam = point_ops.camera_block(c2w, fx, fy, cx, cy, gpu)
    b = point_ops.ray_samples_camera(cam, W, first, depth, radius, 10, 0.95, 1.05)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert int(b[4]) == 1


def test_render_fast_path_equals_general_path(gpu):
    """all rays with a depth prior: render_batch_ray takes the HIP-only path (ray_samples, KNN, gather,
    decoders, ray_counts, compositing); same numbers as the general path built from torch ops.  One ray
    without depth sends the batch back to the general path."""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd import point_ops
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd.renderer import Renderer
    cfg = _cfg(gpu)
    pts, geo, col = synth.box_cloud(n_hits=40000)
    H, W = 24, 32
    ro, rd, depth, radius, c2w = synth.box_rays(H, W, fx=16.0, fy=16.0, cx=15.5, cy=11.5)
    t = lambda x: torch.from_numpy(x).to(gpu)
    npc = NeuralPointCloud(cfg)
    npc.add_points(t(pts), t(geo), t(col))
    torch.manual_seed(43)
    dec = POINT(cfg, use_view_direction=True).eval().to(gpu)

    class Cam:
        H, W, fx, fy, cx, cy = 24, 32, 16.0, 16.0, 15.5, 11.5
    ren = Renderer(cfg, Cam())
    kw = dict(npc_geo_feats=npc.geo_feats, npc_col_feats=npc.col_feats, cloud_pos=npc.cloud_pos(),
              dynamic_r_query=t(radius * 2.5))
    # the sampling kernel reproduces the torch chain bit for bit
    z, p, v, rs, nzero = point_ops.ray_samples(t(ro), t(rd), t(depth), t(radius), 10, 0.95, 1.05)
    tl = torch.linspace(0.0, 1.0, steps=10, device=gpu)
    g = t(depth).reshape(-1, 1)
    z_ref = 0.95 * g * (1. - tl) + 1.05 * g * tl
    assert torch.equal(z, z_ref) and int(nzero) == 0
    assert torch.equal(p, (t(ro)[:, None, :] + t(rd)[:, None, :] * z_ref[:, :, None]).reshape(-1, 3))
    assert torch.equal(v, t(rd).repeat_interleave(10, dim=0)) and torch.equal(rs, t(radius).repeat_interleave(10))
    with torch.no_grad():
        for stage in ("color", "geometry"):
            ren.use_fast_path = True
            fast = ren.render_batch_ray(npc, dec, t(rd), t(ro), gpu, stage, gt_depth=t(depth), **kw)
            ren.use_fast_path = False
            gen = ren.render_batch_ray(npc, dec, t(rd), t(ro), gpu, stage, gt_depth=t(depth), **kw)
            for a, b in zip(fast, gen):
                assert a.dtype == b.dtype and a.shape == b.shape
                if a.dtype.is_floating_point:
                    torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
                else:
                    assert torch.equal(a, b)
        gt0 = depth.copy()
        gt0[5] = 0.0
        ren.use_fast_path = True
        a = ren.render_batch_ray(npc, dec, t(rd), t(ro), gpu, "color", gt_depth=t(gt0), **kw)
        ren.use_fast_path = False
        b = ren.render_batch_ray(npc, dec, t(rd), t(ro), gpu, "color", gt_depth=t(gt0), **kw)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    c, vld = point_ops.ray_counts(torch.tensor([1, 1, 1, 0, 0, 1, 0, 0], dtype=torch.bool, device=gpu), 4, 3)
    assert c.tolist() == [3, 1] and vld.tolist() == [True, False]


def test_fused_mlp_matches_torch_path_and_reference(gpu):
    """the MFMA decoder kernels == the torch nn.Linear path == the reference fixture"""
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    f = np.load(os.path.join(GOLD, "decoders.npz"))
    t = lambda k: torch.from_numpy(f[k]).to(gpu)
    torch.manual_seed(43)
    dec = POINT(_cfg(gpu), c_dim=32, hidden_size=128, use_view_direction=True).eval().to(gpu)
    npc = NeuralPointCloud(_cfg(gpu))
    npc.add_points(t("cloud"), t("geo"), t("col"))
    args = (t("p")[None], npc, "color", npc.geo_feats, npc.col_feats)
    kw = dict(pts_num=10, cloud_pos=npc.cloud_pos(), pts_views_d=t("views"), dynamic_r_query=t("radius"))
    with torch.no_grad():
        dec.use_fused = True
        raw_f, rm_f, pm_f, cnt_f = dec(*args, **kw)
        dec.use_fused = False
        raw_t, rm_t, pm_t, cnt_t = dec(*args, **kw)
    pm = f["point_mask"]
    assert torch.equal(pm_f, pm_t) and torch.equal(rm_f, rm_t) and torch.equal(cnt_f, cnt_t)
    rf, rt = raw_f.cpu().numpy(), raw_t.cpu().numpy()
    assert np.all(rf[~pm, 3] == -100.0)
    np.testing.assert_allclose(rf[pm, 3], rt[pm, 3], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(rf[pm, :3], rt[pm, :3], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(rf[pm, 3], f["occ"][pm], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(rf[pm, :3], f["rgb"][pm], rtol=1e-3, atol=1e-3)
    with torch.no_grad():
        dec.use_fused = True
        raw_g, *_ = dec(args[0], npc, "geometry", npc.geo_feats, npc.col_feats, **kw)
    assert torch.all(raw_g[:, :3] == 0) and torch.allclose(raw_g[:, 3], raw_f[:, 3])


def test_fused_mlp_ragged_sizes(gpu):
    """sample counts that are not multiples of the 64-sample tile"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    cfg = _cfg(gpu)
    pts, geo, col = synth.box_cloud(n_hits=3000)
    t = lambda x: torch.from_numpy(x).to(gpu)
    npc = NeuralPointCloud(cfg)
    npc.add_points(t(pts), t(geo), t(col))
    torch.manual_seed(1)
    dec = POINT(cfg, use_view_direction=True).eval().to(gpu)
    rng = np.random.default_rng(0)
    for Q in (10, 70, 650):
        p = t((pts[rng.integers(0, len(pts), Q)] * rng.uniform(0.97, 1.03, (Q, 1))).astype(np.float32))
        v = t(rng.standard_normal((Q, 3)).astype(np.float32))
        r = t(rng.uniform(0.05, 0.3, (Q, 1)).astype(np.float32))
        with torch.no_grad():
            dec.use_fused = True
            a, _, pm, _ = dec(p[None], npc, "color", npc.geo_feats, npc.col_feats, pts_num=10, pts_views_d=v, dynamic_r_query=r)
            dec.use_fused = False
            b, _, pm2, _ = dec(p[None], npc, "color", npc.geo_feats, npc.col_feats, pts_num=10, pts_views_d=v, dynamic_r_query=r)
        assert torch.equal(pm, pm2)
        b[~pm, 3] = -100.0          # Renderer.py:206-207 (the fused path applies it itself)
        a, b, pm = a.cpu().numpy(), b.cpu().numpy(), pm.cpu().numpy()
        np.testing.assert_allclose(a[:, 3], b[:, 3], rtol=2e-3, atol=1e-3)
        np.testing.assert_allclose(a[pm, :3], b[pm, :3], rtol=2e-3, atol=1e-3)


# ---- point-cloud maintenance ---------------------------------------------------
def _npc_cfg(dev, H=48, W=64):
    cfg = _cfg(dev)
    cfg["cam"] = {"H": H, "W": W, "fx": 40.0, "fy": 40.0, "cx": W / 2 - 0.5, "cy": H / 2 - 0.5, "H_out": H, "W_out": W,
                  "H_edge": 0, "W_edge": 0}
    cfg["pointcloud"].update(N_add=3, near_end_surface=0.98, far_end_surface=1.02, radius_add=0.04, radius_min=0.02,
                             fix_interval_when_add_along_ray=False)
    return cfg


def test_add_neural_points_radius_test_and_deformation(gpu):
    """add_neural_points (neural_point.py:165-262) against a brute-force restatement of its radius test,
    then update_points_pos (378-438) + retrain_updated_points"""
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd.common import get_rays_from_uv
    cfg = _npc_cfg(gpu)
    H, W = 48, 64
    npc = NeuralPointCloud(cfg)
    g = torch.Generator(device="cpu").manual_seed(3)
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([0.1, -0.2, 0.3])
    jj, ii = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    ii, jj = ii.reshape(-1).to(gpu), jj.reshape(-1).to(gpu)
    depth = (1.5 + 0.3 * torch.rand(H, W, generator=g)).to(gpu)
    depth[::7, ::5] = 0.0                                          # holes
    depth[3, 4] = 40.0                                             # beyond 2 x the 0.8 quantile: rejected
    ro, rd = get_rays_from_uv(ii.float(), jj.float(), c2w, 40.0, 40.0, W / 2 - 0.5, H / 2 - 0.5, gpu)
    col = torch.rand(H * W, 3, generator=g).to(gpu)
    d = depth[jj, ii]
    n1 = int(npc.add_neural_points(ro, rd, d, col, 0, ii, jj))
    keep = (d > 0) & (d < d.quantile(0.8) * 2.0)
    assert n1 == int(keep.sum()) and npc.pts_num() == 3 * n1         # empty cloud: every valid ray accepted
    assert npc.input_pos().shape == (n1, 3) and int(npc.input_video_idx().max()) == 0
    # second keyframe, shifted camera: only rays whose surface point has no neighbour within radius_add
    c2w2 = c2w.clone()
    c2w2[0, 3] += 0.02
    ro2, rd2 = get_rays_from_uv(ii.float(), jj.float(), c2w2, 40.0, 40.0, W / 2 - 0.5, H / 2 - 0.5, gpu)
    cloud_before = npc.cloud_pos().clone()
    n2 = int(npc.add_neural_points(ro2, rd2, d, col, 1, ii, jj))
    pts_gt = (ro2 + rd2 * d[:, None])[keep]
    dist = torch.cdist(pts_gt.double(), cloud_before.double())
    expect = (dist.min(1).values ** 2 > 0.04 ** 2)
    near_tie = ((dist.min(1).values ** 2 - 0.04 ** 2).abs() < 1e-6)
    assert abs(n2 - int(expect.sum())) <= int(near_tie.sum())
    assert npc.pts_num() == 3 * (n1 + n2) and npc.index.ntotal == npc.pts_num()
    assert npc.geo_feats.shape == (npc.pts_num(), cfg["model"]["c_dim"])
    # deformation of keyframe 0: new depth map (with holes -> rescaled previous depth), same pose
    depth2 = depth * 1.1
    depth2[5, 6] = 0.0
    before = npc.cloud_pos().clone()
    npc.update_points_pos(0, depth2, c2w, cfg)
    npc.retrain_updated_points()
    moved = (npc.cloud_pos() - before).abs().sum(1) > 0
    assert int(moved.sum()) >= 3 * n1 - 9 and not bool(moved[3 * n1:].any())     # only keyframe 0's points moved
    fm = npc.input_video_idx() == 0
    z = ((npc.input_pos()[fm] - c2w[:3, 3].to(gpu)) * -1)[:, 2]                 # camera looks down -z
    torch.testing.assert_close(z, npc._input_depth[fm], atol=1e-5, rtol=1e-5)
    D, I, nn = npc.find_neighbors_faiss(npc.input_pos()[:64], step="query")
    assert bool((nn > 0).all())


def test_add_points_from_video_unprojects_keyframes(gpu):
    """add_points(video_idxs) (neural_point.py:145-162): iproj with the inverse stored pose"""
    from test_gpu_graph import make_video
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from oracle import se3 as ose3
    g, video = make_video(gpu, 5, 6, 8)
    n = video.counter.value
    video.disps_up[:n] = torch.nn.functional.interpolate(video.disps[:n, None], scale_factor=8, mode="nearest")[:, 0]
    video.valid_depth_mask[:n] = True
    video.valid_depth_mask[1, :4] = False
    npc = NeuralPointCloud(_npc_cfg(gpu), video)
    cnt = int(npc.add_points(torch.tensor([1, 3], device=gpu)))
    assert cnt == int(video.valid_depth_mask[[1, 3]].sum())
    assert torch.equal(npc.full_mask()[1], video.valid_depth_mask[1])
    # pixel (v,u) of keyframe 3: X_world = inv(w2c) * ((u-cx)/fx, (v-cy)/fy, 1) / disp
    fx, fy, cx, cy = (video.intrinsics[0] * 8).cpu().numpy()
    v, u = 20, 33
    dsp = float(video.disps_up[3, v, u])
    Xc = np.array([(u - cx) / fx / dsp, (v - cy) / fy / dsp, 1.0 / dsp, 1.0])
    Xw = np.linalg.inv(ose3.matrix(g["poses"][3])) @ Xc
    np.testing.assert_allclose(npc.full_pcl()[3, v, u].cpu().numpy(), Xw[:3], rtol=1e-4, atol=1e-4)


def test_sample_near_pcl_matches_host_loop(gpu):
    """R6 (neural_point.py:315-375): samples of depth-less rays between their first two occupied probes, all
    rays at once on the device == the per-ray numpy loop of the reference formulation"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.neural_point import NeuralPointCloud
    cfg = _cfg(gpu)
    pts, geo, col = synth.box_cloud(n_hits=40000)
    npc = NeuralPointCloud(cfg)
    npc.add_points(torch.from_numpy(pts).to(gpu), torch.from_numpy(geo).to(gpu), torch.from_numpy(col).to(gpu))
    ro, rd, depth, radius, c2w = synth.box_rays(24, 32, fx=16.0, fy=16.0, cx=15.5, cy=11.5)
    ro, rd = torch.from_numpy(ro).to(gpu), torch.from_numpy(rd).to(gpu)
    rd[:7] = -rd[:7] * 0 + torch.tensor([0.0, 0.0, 1e-3], device=gpu)     # rays that never reach the cloud
    near, far, num, intervals = 0.3, 6.0, 10, 25
    z, invalid = npc.sample_near_pcl(ro, rd, near, far, num)
    assert z.shape == (ro.shape[0], num) and z.dtype == torch.float32
    # host restatement
    z_probe = torch.linspace(near, far, steps=intervals, device=gpu)
    p = (ro[:, None, :] + rd[:, None, :] * z_probe[None, :, None]).reshape(-1, 3)
    _, _, nn_num = npc.find_neighbors_faiss(p, step="query")
    occ = (nn_num.reshape(-1, intervals) > 0).cpu().numpy()
    z_section = np.linspace(near, far, intervals)
    ref = np.tile(np.linspace(near, far, num), (occ.shape[0], 1))
    inv_ref = occ.sum(1) < 2
    for r in np.nonzero(~inv_ref)[0]:
        c = np.nonzero(occ[r])[0]
        ref[r] = np.linspace(z_section[c[0]], z_section[c[1]], num=num)
    assert np.array_equal(invalid.cpu().numpy(), inv_ref) and inv_ref[:7].all() and (~inv_ref).sum() > 100
    np.testing.assert_allclose(z.cpu().numpy(), ref.astype(np.float32), rtol=0, atol=1e-6)


def test_proxy_depth_projection_and_deformation_driver(gpu):
    """proj_depth_map (neural_point.py:446-506) against a scatter-min restatement of the same projection;
    get_proxy_render_depth (:539-575) fills holes of the tracker depth from the projected cloud, then from the
    mono prior; update_points_pos(npc, video) (:509-537) consumes the npc_dirty flags"""
    from test_gpu_graph import make_video
    from glorie_slam_amd import neural_point as NP
    H, W = 48, 64
    cfg = _npc_cfg(gpu, H, W)
    cfg["mapping"] = {"mapping_window_size": 5, "render_depth": "proxy", "save_depth": False}
    npc = NP.NeuralPointCloud(cfg)
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-2, 2, (20000, 2)), rng.uniform(-4, -1, (20000, 1))], 1).astype(np.float32)
    pts[:50, 2] = 1.0                                           # behind the camera: never projected
    npc.add_points(torch.from_numpy(pts).to(gpu))
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([0.05, -0.02, 0.1])
    ang = 0.1
    c2w[:3, :3] = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    proj = NP.proj_depth_map(c2w, npc, gpu, cfg, neural_pcl=True)
    # restatement: X_c = w2c X, x flipped, (u, v) = trunc(K X_c / z), depth = -z, per-pixel minimum
    w2c = torch.linalg.inv(c2w.double())
    X = torch.from_numpy(pts).double()
    Xc = (w2c[:3, :3] @ X.T).T + w2c[:3, 3]
    z = Xc[:, 2] + 1e-6
    u = (40.0 * -Xc[:, 0] + (W / 2 - 0.5) * Xc[:, 2]) / z
    v = (40.0 * Xc[:, 1] + (H / 2 - 0.5) * Xc[:, 2]) / z
    m = (u < W) & (u >= 0) & (v < H) & (v >= 0) & (-z > 0)
    ref = torch.full((H * W,), float("inf"), dtype=torch.float64)
    ref.scatter_reduce_(0, (v[m].long() * W + u[m].long()), -z[m], reduce="amin")
    ref = torch.where(torch.isinf(ref), torch.zeros_like(ref), ref).reshape(H, W)
    got = proj.cpu().double()
    differ = (got - ref).abs() > 1e-4 * (1 + ref.abs())
    assert differ.float().mean() < 2e-3                         # fp32 vs fp64 pixel assignment at cell borders
    assert (got > 0).float().mean() > 0.5 and float(got.max()) < 4.2
    # proxy depth: tracker depth wins, then the projection, then the mono prior
    class V:
        pass
    droid = torch.zeros(H, W, device=gpu)
    droid[:, : W // 2] = 2.5
    mono = torch.full((H, W), 7.0, device=gpu)
    npc.video = V()
    npc.video.counter = type("C", (), {"value": 3})()
    npc._full_pcl = torch.from_numpy(pts[:H * W * 2]).to(gpu).reshape(2, H, W, 3).contiguous()
    npc._full_pcl = torch.cat([npc._full_pcl, torch.zeros(6, H, W, 3, device=gpu)])
    npc._full_mask = torch.zeros(8, H, W, dtype=torch.bool, device=gpu)
    npc._full_mask[:2] = True
    proxy = NP.get_proxy_render_depth(npc, cfg, c2w, droid, mono, gpu)
    p2 = NP.proj_depth_map(c2w, npc, gpu, cfg)
    assert torch.equal(proxy[:, : W // 2], droid[:, : W // 2])
    right = proxy[:, W // 2:]
    exp = torch.where(p2[:, W // 2:] > 0, p2[:, W // 2:], mono[:, W // 2:])
    assert torch.equal(right, exp) and bool((right == 7.0).any()) and bool((right < 7.0).any())
    # the frame mapping_window_size behind the newest is left out: counter 3 -> index -2 = frame 6 (empty): same map;
    # counter 6 -> frame 1 dropped: fewer hits
    npc.video.counter.value = 6
    p3 = NP.proj_depth_map(c2w, npc, gpu, cfg)
    assert int((p3 > 0).sum()) < int((p2 > 0).sum())

    # deformation driver on a real video
    g, video = make_video(gpu, 5, 6, 8)
    n = video.counter.value
    video.cfg["mapping"] = {"render_depth": "proxy", "mapping_window_size": 5}
    video.cfg["cam"] = _npc_cfg(gpu, 48, 64)["cam"]
    video.disps_up[:n] = torch.nn.functional.interpolate(video.disps[:n, None], scale_factor=8, mode="nearest")[:, 0]
    video.valid_depth_mask[:n] = True
    npc2 = NP.NeuralPointCloud(_npc_cfg(gpu, 48, 64), video)
    jj, ii = torch.meshgrid(torch.arange(48), torch.arange(64), indexing="ij")
    ii, jj = ii.reshape(-1).to(gpu), jj.reshape(-1).to(gpu)
    from glorie_slam_amd.common import get_rays_from_uv
    est_depth, est_mask, c2w1 = video.get_depth_and_pose(1, gpu)
    c2w1[:3, 1:3] *= -1
    ro, rd = get_rays_from_uv(ii.float(), jj.float(), c2w1, 40.0, 40.0, 31.5, 23.5, gpu)
    npc2.add_neural_points(ro, rd, est_depth[jj, ii], torch.rand(48 * 64, 3, device=gpu), 1, ii, jj)
    before = npc2.cloud_pos().clone()
    video.npc_dirty[:] = False
    NP.update_points_pos(npc2, video)                             # nothing dirty: no-op
    assert torch.equal(npc2.cloud_pos(), before)
    video.disps_up[1] *= 0.9                                      # the tracker revised keyframe 1
    video.npc_dirty[1] = True
    NP.update_points_pos(npc2, video)
    assert not bool(video.npc_dirty.any())
    assert float((npc2.cloud_pos() - before).abs().max()) > 1e-3 and npc2.index.ntotal == npc2.pts_num()
    assert bool(npc2.full_mask()[1].all())


def test_full_frame_render_is_order_independent(gpu):
    """BASELINE size (307,200 rays x 10 samples against the 524k-point cloud): a ray's result does not depend on
    where in which batch it is evaluated - rendering a random permutation of the rays and un-permuting gives
    the frame bit for bit (exact KNN with a total order, per-sample decoders, per-ray compositing) - and two
    passes over the same rays are identical."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    npc, dec, ren, rays = bench.build_renderer(gpu)

    def frame(order=None):
        outs = []
        o, d, dep, rad = (rays[k] if order is None else rays[k][order] for k in ("o", "d", "depth", "radius"))
        bs = ren.ray_batch_size
        with torch.no_grad():
            for i in range(0, o.shape[0], bs):
                outs.append(ren.render_batch_ray(npc, dec, d[i:i + bs], o[i:i + bs], gpu, "color",
                                                 gt_depth=dep[i:i + bs], npc_geo_feats=npc.geo_feats,
                                                 npc_col_feats=npc.col_feats, cloud_pos=npc.cloud_pos(),
                                                 dynamic_r_query=rad[i:i + bs]))
        return [torch.cat([b[k] for b in outs]) for k in range(5)]

    a, b = frame(), frame()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    perm = torch.randperm(rays["o"].shape[0], device=gpu, generator=torch.Generator(device=gpu).manual_seed(3))
    c = frame(perm)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=gpu)
    for x, y in zip(a, c):
        assert torch.equal(x, y[inv])
    depth, unc, color, valid, counts = a
    assert valid.float().mean() > 0.99 and torch.isfinite(depth).all() and torch.isfinite(color).all()
    true_depth = rays["depth"]
    assert float(((depth - true_depth).abs() / true_depth)[valid].max()) < 0.06      # samples span 0.95 .. 1.05 d


def test_knn_bit_exact_on_the_baseline_cloud(gpu):
    """BASELINE size: the 524,288-point cloud of the bench (cell size derived from its bounding box, the max_cells clamp,
    multi-shell walks) searched the way the renderer searches it - the samples of 48 image rows in patch order, per-query
    radii, weights and mask from the same launch, stopping at the query ball - and 4,608 of those queries, spread over the
    strips, checked against brute force over all points: indices and squared distances bit for bit (inside the ball for the
    bounded search, all 8 slots for the plain one), neighbour counts, weights."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from glorie_slam_amd import point_ops
    npc, dec, ren, rays = bench.build_renderer(gpu)
    pts_host = npc.cloud_pos().cpu().numpy()
    assert pts_host.shape[0] == 524286                           # 174,762 hits x 3 (Np = 524,288 nominal)
    S, W = ren.N_surface, 640
    rng = np.random.default_rng(11)
    checked = 0
    for row0 in (0, 232, 464):                                   # top strip (image border), middle, bottom
        sl = slice(row0 * W, (row0 + 16) * W)
        z, q, views, rq, nz = point_ops.ray_samples(rays["o"][sl], rays["d"][sl], rays["depth"][sl], rays["radius"][sl], S,
                                                    ren.near_end_surface, ren.far_end_surface)
        D, I, nn, w, has = npc.index.search(q, 8, radius_per_query=rq, image_layout=(S, W), weights=(2, False, True))
        D2, I2, nn2 = npc.index.search(q, 8, radius_per_query=rq)                      # plain exact search, any order
        pick = np.sort(rng.choice(q.shape[0], 1536, replace=False))
        qh, rh = q[pick].cpu().numpy(), rq[pick].cpu().numpy()
        rD, rI = oknn.knn_bruteforce(pts_host, qh, 8, chunk=256)
        assert np.array_equal(I2[pick].cpu().numpy(), rI) and np.array_equal(D2[pick].cpu().numpy(), rD)
        rnn = oknn.neighbor_count(rD, rh)
        assert np.array_equal(nn2[pick].cpu().numpy(), rnn) and np.array_equal(nn[pick].cpu().numpy(), rnn)
        inside = rD < (rh * rh)[:, None]
        Db, Ib = D[pick].cpu().numpy(), I[pick].cpu().numpy()
        assert np.array_equal(Ib[inside], rI[inside]) and np.array_equal(Db[inside], rD[inside])
        # the weights of the same launch: 1 / (D + 1e-10) inside the ball, L1-normalised, zero outside; mask = at least 2 inside
        wr = np.where(inside, 1.0 / (rD.astype(np.float32) + np.float32(1e-10)), 0.0).astype(np.float32)
        wr = wr / np.maximum(wr.sum(1, keepdims=True), np.float32(1e-30))
        ok = rnn >= 2
        np.testing.assert_allclose(w[pick].cpu().numpy()[ok], wr[ok], rtol=2e-6, atol=1e-7)
        assert np.array_equal(has[pick].cpu().numpy().astype(bool), ok)
        assert 3.0 < rnn.mean() <= 8.0                            # the radii of the bench give partially filled balls
        checked += len(pick)
    assert checked == 4608
