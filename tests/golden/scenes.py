"""Seeded INPUTS of the reference-minted fixtures (render.npz, render_grad.npz): configuration dicts and the wall scene.
Data only - this module knows nothing about /root/reference; tests/golden/make_golden.py (which imports the reference, in
the build container only) and the GPU tests both take the inputs from here, so the GPU suite never imports the minting
script."""
import torch


def decoder_cfg():
    return {"pointcloud": {"nn_weighting": "distance", "use_dynamic_radius": True, "min_nn_num": 2,
                           "nn_num": 8, "radius_query": 0.08},
            "rendering": {"N_surface": 10},
            "model": {"encode_rel_pos_in_col": True, "encode_viewd": True, "c_dim": 32}}


def render_cfg():
    cfg = decoder_cfg()
    cfg["rendering"] = {"N_surface": 10, "near_end_surface": 0.95, "far_end_surface": 1.05,
                        "sample_near_pcl": True, "sigmoid_coef": 0.1, "near_end": 0.3}
    return cfg


def render_scene():
    """a wall at x = 2 m seen from the origin (camera looking along +x, OpenGL rays of get_rays):
    4000 surface hits x 3 along-ray copies (N_add = 3), 12x16 image whose top / bottom rows miss the wall"""
    g = torch.Generator().manual_seed(11)
    hits = torch.stack([torch.full((4000,), 2.0), torch.rand(4000, generator=g) * 3.0 - 1.5,
                        torch.rand(4000, generator=g) * 2.0 - 1.0], -1)
    cloud = torch.cat([hits * s for s in (0.95, 1.0, 1.05)], 0) + 0.005 * torch.randn(12000, 3, generator=g)
    geo = torch.randn(12000, 32, generator=g) * 0.1
    col = torch.randn(12000, 32, generator=g) * 0.1
    c2w = torch.eye(4)
    c2w[:3, :3] = torch.tensor([[0.0, 0.0, -1.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    cam = dict(H=12, W=16, fx=10.0, fy=10.0, cx=7.5, cy=5.5)
    depth = 2.0 * (1.0 + 0.01 * torch.randn(12 * 16, generator=g))
    radius = (torch.rand(12 * 16, generator=g) * 0.12 + 0.04) * depth / 3.0
    depth_zero = depth.clone()
    depth_zero[torch.tensor([5, 40, 41, 77, 100, 150, 191])] = 0.0         # rays without a depth prior
    return cloud, geo, col, c2w, cam, depth, depth_zero, radius


