"""Golden fixtures minted from the reference's importable Python (tests/golden/make_golden.py)
pin (a) the oracle restatements and (b) the host-side mirrors of the reference modules.
Everything here runs on CPU."""
import os

import numpy as np
import torch

from oracle import corr as ocorr, geom as ogeom

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def test_corr_pyramid_oracle_and_corrblock():
    f = gold("corr_pyramid.npz")
    lv = ocorr.corr_pyramid_fp32(f["fmap1"][0], f["fmap2"][0], num_levels=3)
    for i in range(3):
        np.testing.assert_allclose(lv[i], f[f"level{i}"], rtol=1e-5, atol=1e-5)
    from glorie_slam_amd.droid_net import CorrBlock
    cb = CorrBlock(torch.from_numpy(f["fmap1"]), torch.from_numpy(f["fmap2"]), num_levels=3)
    for i in range(3):
        np.testing.assert_allclose(cb.corr_pyramid[i].numpy(), f[f"level{i}"], rtol=1e-5, atol=1e-5)


def test_cvx_upsample_oracle():
    f = gold("cvx_upsample.npz")
    up = ogeom.cvx_upsample(f["data"][..., 0], f["mask"])
    np.testing.assert_allclose(up, f["up"][..., 0], rtol=1e-5, atol=1e-6)


def test_update_module_same_init_and_outputs():
    from glorie_slam_amd.droid_net import UpdateModule
    f = gold("update_module.npz")
    torch.manual_seed(43)
    net = UpdateModule().eval()
    assert sum(p.numel() for p in net.parameters()) == int(f["nparams"]) == 2556933
    psum = float(sum(p.detach().double().abs().sum() for p in net.parameters()))
    assert abs(psum - float(f["param_abs_sum"])) < 1e-6 * psum  # same RNG consumption order
    fi = gold("update_module_inputs.npz")          # the fixture's inputs as minted (not re-derived from a torch seed)
    N, h, w = 3, 8, 10
    x_net, x_inp, x_corr, x_flow = (torch.from_numpy(fi[k]) for k in ("net", "inp", "corr", "flow"))
    assert x_net.shape == (1, N, 128, h, w) and x_corr.shape == (1, N, 196, h, w)
    with torch.no_grad():
        o = net(x_net, x_inp, x_corr, x_flow, torch.from_numpy(f["ii"]), torch.from_numpy(f["jj"]))
        gru = net.gru(x_net[0], x_inp[0], x_corr[0, :, :128], x_flow[0, :, :1].repeat(1, 64, 1, 1))
    for got, key, tol in zip(o, ("net", "delta", "weight", "eta", "upmask"), (1e-5, 1e-5, 1e-5, 1e-6, 2e-3)):
        np.testing.assert_allclose(got.numpy(), f[key].astype(np.float32), rtol=tol, atol=tol)
    np.testing.assert_allclose(gru.numpy(), f["gru"], rtol=1e-5, atol=1e-5)
    assert o[3].shape == (1, 2, h, w) and o[4].shape == (1, 2, 576, h, w)  # 2 distinct source frames


def test_state_dict_names_match_reference():
    """droid.pth loading contract (slam.py:70-81)"""
    from glorie_slam_amd.droid_net import UpdateModule
    keys = set(UpdateModule().state_dict().keys())
    for k in ("weight.2.weight", "weight.2.bias", "delta.2.weight", "delta.2.bias", "gru.convz.weight",
              "gru.convq_glo.bias", "agg.eta.0.weight", "agg.upmask.0.weight", "corr_encoder.0.weight",
              "flow_encoder.2.bias"):
        assert k in keys


def test_compositing_torch_path():
    from glorie_slam_amd.common import raw2outputs_nerf_color
    f = gold("raw2outputs.npz")
    d, v, c, w = raw2outputs_nerf_color(torch.from_numpy(f["raw"]).clone(), torch.from_numpy(f["z"]),
                                        torch.from_numpy(f["rays_d"]), device="cpu", coef=0.1)
    for got, key in ((d, "depth"), (v, "var"), (c, "rgb"), (w, "weights")):
        np.testing.assert_allclose(got.numpy(), f[key], rtol=1e-5, atol=1e-6)


def test_align_and_rays():
    from glorie_slam_amd.common import align_scale_and_shift, get_rays, get_rays_from_uv
    f = gold("align.npz")
    s, q, e = align_scale_and_shift(torch.from_numpy(f["pred"]), torch.from_numpy(f["tgt"]), torch.from_numpy(f["wts"]))
    np.testing.assert_allclose(s.numpy(), f["scale"], rtol=1e-5)
    np.testing.assert_allclose(q.numpy(), f["shift"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(e.numpy(), f["err"], rtol=1e-4)
    f = gold("rays.npz")
    ro, rd = get_rays(6, 8, 10.0, 11.0, 3.5, 2.5, torch.from_numpy(f["c2w"]), "cpu")
    np.testing.assert_allclose(ro.numpy(), f["rays_o"], atol=1e-6)
    np.testing.assert_allclose(rd.numpy(), f["rays_d"], atol=1e-6)
    uo, ud = get_rays_from_uv(torch.tensor([1.0, 5.0]), torch.tensor([2.0, 0.0]), torch.from_numpy(f["c2w"]),
                              10.0, 11.0, 3.5, 2.5, "cpu")
    np.testing.assert_allclose(ud.numpy(), f["uv_d"], atol=1e-6)


def _decoder_cfg():
    return {"pointcloud": {"nn_weighting": "distance", "use_dynamic_radius": True, "min_nn_num": 2,
                           "nn_num": 8, "radius_query": 0.08},
            "rendering": {"N_surface": 10},
            "model": {"encode_rel_pos_in_col": True, "encode_viewd": True, "c_dim": 32}}


class BruteNPC:
    def __init__(self, pos):
        self.pos = pos

    def get_radius_query(self):
        return 0.08

    def cloud_pos(self, index=None):
        return self.pos

    def find_neighbors_faiss(self, p, step='query', retrain=False, is_pts_grad=False, dynamic_radius=None):
        from oracle import knn as oknn
        D, I = oknn.knn_bruteforce(self.pos.numpy(), p.numpy(), 8)
        D, I = torch.from_numpy(D), torch.from_numpy(I)
        r2 = dynamic_radius.reshape(-1, 1) ** 2
        return D, I, (D < r2).sum(-1).int()


def test_decoders_match_reference():
    from glorie_slam_amd.decoder import POINT
    f = gold("decoders.npz")
    torch.manual_seed(43)
    dec = POINT(_decoder_cfg(), c_dim=32, hidden_size=128, use_view_direction=True).eval()
    sd = {k[4:]: torch.from_numpy(f[k]) for k in f.files if k.startswith("sd__")}
    mine = dec.state_dict()
    assert set(mine.keys()) == set(sd.keys())
    for k in sd:  # identical default initialisation under the same seed
        np.testing.assert_allclose(mine[k].numpy(), sd[k].numpy(), rtol=0, atol=0, err_msg=k)
    np.testing.assert_array_equal(dec.color_decoder.embedder._B.numpy(), f["color_B_pos"])
    np.testing.assert_array_equal(dec.color_decoder.embedder_view_direction._B.numpy(), f["color_B_view"])
    npc = BruteNPC(torch.from_numpy(f["cloud"]))
    p = torch.from_numpy(f["p"])
    rad = torch.from_numpy(f["radius"])
    with torch.no_grad():
        raw, ray_mask, point_mask, counter = dec(p[None], npc, "color", torch.from_numpy(f["geo"]),
                                                 torch.from_numpy(f["col"]), pts_num=10,
                                                 cloud_pos=torch.from_numpy(f["cloud"]),
                                                 pts_views_d=torch.from_numpy(f["views"]), dynamic_r_query=rad)
    pm = f["point_mask"]
    assert np.array_equal(point_mask.numpy(), pm) and pm.sum() > 50 and (~pm).sum() >= 10
    assert np.array_equal(ray_mask.numpy(), f["ray_mask"])
    assert np.array_equal(counter.numpy(), f["counter"])
    # samples without neighbours get a random feature in the reference -> excluded
    np.testing.assert_allclose(raw[pm, 3].numpy(), f["occ"][pm], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(raw[pm, :3].numpy(), f["rgb"][pm], rtol=1e-4, atol=1e-5)


def test_schur_solve_oracle_matches_reference():
    from oracle import dspo as odspo
    f = gold("schur_solve.npz")
    dx, dz = odspo.schur_solve(f["H"][0], f["E"][0], f["C"][0], f["v"][0], f["w"][0], 0.1, 1e-4)
    np.testing.assert_allclose(dx, f["dx"][0], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(dz, f["dz"][0], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(odspo.block_solve(f["H"][0], f["v"][0]), f["block_x"][0], rtol=2e-4, atol=1e-5)
    dxb, dzb = odspo.schur_solve(-f["H"][0], f["E"][0], f["C"][0], f["v"][0], f["w"][0], 0.1, 1e-4)
    assert np.all(f["dx_bad"] == 0) and np.all(dxb == 0)      # non-PD -> zero update
    np.testing.assert_allclose(dzb, f["dz_bad"][0], rtol=2e-4, atol=1e-5)


def test_droidnet_matches_reference():
    """DroidNet mirror (droid.pth loading contract of slam.py:70-81): state-dict names and shapes, identical default
    initialisation under the seed, encoder outputs == the reference's on the same image pair (fixture F13)"""
    from glorie_slam_amd.droid_net import DroidNet
    f = gold("droidnet.npz")
    torch.manual_seed(43)
    net = DroidNet().eval()
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in f["keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(v) for v in f["shapes"]]
    psum = sum(float(v.double().abs().sum()) for v in sd.values())
    assert abs(psum - float(f["param_abs_sum"])) < 1e-9 * psum
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 2, 3, 48, 64, generator=g)
    with torch.no_grad():
        np.testing.assert_allclose(net.fnet(x).numpy(), f["fmap"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(net.cnet(x).numpy(), f["cmap"], rtol=1e-5, atol=1e-5)


def test_se3_helper_matches_oracle_conventions():
    """glorie_slam_amd.lie.SE3 (the lietorch subset of the drivers) against oracle/se3.py and the matrix group"""
    from glorie_slam_amd.lie import SE3
    from oracle import se3 as ose3
    torch.manual_seed(0)
    xi = torch.randn(64, 6) * torch.tensor([1, 1, 1, 0.8, 0.8, 0.8])
    xi[:5, 3:] *= 1e-6                                  # the small-angle branches
    G = SE3.exp(xi)
    t, q = ose3.se3_exp(xi.numpy())
    np.testing.assert_allclose(G.data.numpy(), np.concatenate([t, q], -1), atol=1e-6)
    np.testing.assert_allclose(G.log().numpy(), xi.numpy(), atol=5e-6)
    H = SE3.exp(torch.randn(64, 6) * 0.3)
    np.testing.assert_allclose((G * H).matrix().numpy(), (G.matrix() @ H.matrix()).numpy(), atol=1e-6)
    np.testing.assert_allclose((G * G.inv()).matrix().numpy(), np.broadcast_to(np.eye(4), (64, 4, 4)), atol=1e-6)
    for k in range(4):
        np.testing.assert_allclose(G.matrix()[k].numpy(), ose3.matrix(G.data[k].numpy()), atol=1e-6)
    # left retraction of the BA kernels == exp(xi) * G
    r = ose3.retract(xi[7].numpy(), H.data[7].numpy())
    np.testing.assert_allclose((SE3.exp(xi[7:8]) * H[7:8]).data[0].numpy(), r, atol=1e-6)
