"""Ray / compositing / alignment helpers of the hot path -- mirror of the four functions of
/root/reference/src/utils/common.py that SURVEY.md section 2 (#16) puts in scope."""
import numpy as np
import torch

from . import point_ops


def get_rays(H, W, fx, fy, cx, cy, c2w, device, crop_edge_h=0, crop_edge_w=0, return_ij=False):
    """common.py:302-322: pinhole rays of a whole image (OpenGL convention: -y, -z)"""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    c2w = c2w.to(device)
    i, j = torch.meshgrid(torch.linspace(crop_edge_w, W - 1 - crop_edge_w, W - 2 * crop_edge_w),
                          torch.linspace(crop_edge_h, H - 1 - crop_edge_h, H - 2 * crop_edge_h), indexing='ij')
    i, j = i.t().to(device), j.t().to(device)
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    dirs = dirs.reshape(H - 2 * crop_edge_h, W - 2 * crop_edge_w, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    if return_ij:
        return rays_o, rays_d, i.to(torch.int64), j.to(torch.int64)
    return rays_o, rays_d


def update_cam(cfg):
    """common.py:377-398: intrinsics after the resize / edge crop of the pre-processing"""
    cam = cfg['cam']
    H, W = cam['H'], cam['W']
    fx, fy, cx, cy = cam['fx'], cam['fy'], cam['cx'], cam['cy']
    h_edge, w_edge = cam.get('H_edge', 0), cam.get('W_edge', 0)
    H_out, W_out = cam['H_out'], cam['W_out']
    fx = fx * (W_out + w_edge * 2) / W
    fy = fy * (H_out + h_edge * 2) / H
    cx = cx * (W_out + w_edge * 2) / W - w_edge
    cy = cy * (H_out + h_edge * 2) / H - h_edge
    return H_out, W_out, fx, fy, cx, cy


def get_rays_from_uv(i, j, c2w, fx, fy, cx, cy, device):
    """common.py:39-54"""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    c2w = c2w.to(device)
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).to(device).reshape(-1, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    return c2w[:3, -1].expand(rays_d.shape), rays_d


def raw2outputs_nerf_color(raw, z_vals, rays_d, device='cuda:0', coef=0.1):
    """common.py:261-299.  Inference (no grad) goes through the fused HIP compositing kernel;
    with autograd enabled the differentiable torch formulation is used.  Like the reference,
    raw[..., -1] is overwritten with alpha."""
    if not (raw.requires_grad or z_vals.requires_grad) and raw.is_cuda:
        depth, var, rgb, w = point_ops.composite(raw, z_vals, coef)
        raw[..., -1] = torch.sigmoid(coef * raw[..., -1])
        return depth, var, rgb, w
    rgb = raw[..., :-1]
    alpha = torch.sigmoid(coef * raw[..., -1])
    ones = torch.ones((alpha.shape[0], 1), device=alpha.device)
    weights = alpha * torch.cumprod(torch.cat([ones, 1. - alpha + 1e-10], -1), dim=-1)[:, :-1]
    wsum = torch.sum(weights, dim=-1, keepdim=True) + 1e-10
    rgb_map = torch.sum(weights[..., None] * rgb, -2) / wsum
    depth_map = torch.sum(weights * z_vals, -1) / wsum.squeeze(-1)
    tmp = z_vals - depth_map.unsqueeze(-1)
    return depth_map, torch.sum(weights * tmp * tmp, dim=1), rgb_map, weights


@torch.no_grad()
def align_scale_and_shift(prediction, target, weights):
    """common.py:401-437: per-frame weighted least squares  target ~ scale * prediction + shift"""
    if weights is None:
        weights = torch.ones_like(prediction)
    if prediction.dim() < 3:
        prediction, target, weights = prediction[None], target[None], weights[None]
    weights = weights.to(prediction.dtype)
    a00 = torch.sum(weights * prediction * prediction, dim=[1, 2])
    a01 = torch.sum(weights * prediction, dim=[1, 2])
    a11 = torch.sum(weights, dim=[1, 2])
    b0 = torch.sum(weights * prediction * target, dim=[1, 2])
    b1 = torch.sum(weights * target, dim=[1, 2])
    det = a00 * a11 - a01 * a01
    scale = (a11 * b0 - a01 * b1) / det
    shift = (-a01 * b0 + a00 * b1) / det
    err = (scale[:, None, None] * prediction + shift[:, None, None] - target).abs()
    avg = (err * weights).sum(dim=[1, 2]) / weights.sum(dim=[1, 2])
    return scale, shift, avg
