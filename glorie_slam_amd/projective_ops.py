"""Mirror of /root/reference/src/geom/projective_ops.py: same names, arguments and return
values.  `projective_transform(jacobian=False)` is ONE HIP launch (glorie_reproject, what DepthVideo.reproject and
FactorGraph.update run); with `jacobian=True` it additionally returns the reference's (Ji, Jj, Jz) tensors, formed with
torch on the device - the product's DSPO stage never materialises them (csrc/dspo.hip computes Jz per pixel in
registers, csrc/ba.hip the pose Jacobians per edge), so this branch exists for callers and tests that want the
reference's tensors, not for speed.  `poses` is a `lie.SE3` with data [1,B,7] (lietorch.SE3 in the reference)."""
import torch

from . import droid_backends
from .lie import SE3

MIN_DEPTH = 0.2


def extract_intrinsics(intrinsics):
    return intrinsics[..., None, None, :].unbind(dim=-1)


def coords_grid(ht, wd, device):
    y, x = torch.meshgrid(torch.arange(ht).to(device).float(), torch.arange(wd).to(device).float(), indexing="ij")
    return torch.stack([x, y], dim=-1)


def iproj(disps, intrinsics, jacobian=False):
    """pinhole back-projection to homogeneous points (X, Y, 1, disparity), projective_ops.py:18-37"""
    ht, wd = disps.shape[2:]
    fx, fy, cx, cy = extract_intrinsics(intrinsics)
    y, x = torch.meshgrid(torch.arange(ht).to(disps.device).float(), torch.arange(wd).to(disps.device).float(),
                          indexing="ij")
    pts = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(disps), disps], dim=-1)
    if jacobian:
        J = torch.zeros_like(pts)
        J[..., -1] = 1.0
        return pts, J
    return pts, None


def projective_transform(poses, depths, intrinsics, ii, jj, jacobian=False, return_depth=False):
    """map the pixels of frames ii into frames jj (projective_ops.py:96-125): (coords [1,N,h,w,2], valid [1,N,h,w,1])
    and with jacobian=True also (Ji, Jj [1,N,h,w,2,6], Jz [1,N,h,w,2,1])"""
    data = poses.data if isinstance(poses, SE3) else poses
    if return_depth:
        raise NotImplementedError("projective_transform(return_depth=True) has no caller in the reference")
    B = data.shape[0]
    if B != 1:
        raise RuntimeError("projective_transform: batch size 1 (the reference never uses another)")
    ii = ii.to(data.device).long().contiguous()
    jj = jj.to(data.device).long().contiguous()
    intr = intrinsics[0].contiguous().float()
    coords, valid = droid_backends.reproject(data[0].contiguous().float(), depths[0].contiguous().float(), intr, ii, jj)
    coords, valid = coords[None], valid[None]
    if not jacobian:
        return coords, valid
    X0, Jz0 = iproj(depths[:, ii], intrinsics[:, ii], jacobian=True)
    P = SE3(data)
    Gij = P[:, jj] * P[:, ii].inv()
    stereo = torch.tensor([-0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0], device=data.device, dtype=data.dtype)
    Gij = SE3(torch.where((ii == jj)[None, :, None], stereo, Gij.data))
    G4 = SE3(Gij.data[:, :, None, None])
    X1 = G4.act(X0)
    X, Y, Z, d = X1.unbind(dim=-1)
    o = torch.zeros_like(d)
    _, N, H, W = d.shape
    Ja = torch.stack([d, o, o, o, Z, -Y,
                      o, d, o, -Z, o, X,
                      o, o, d, Y, -X, o,
                      o, o, o, o, o, o], dim=-1).view(1, N, H, W, 4, 6)
    fx, fy, cx, cy = extract_intrinsics(intrinsics[:, jj])
    Zc = torch.where(Z < 0.5 * MIN_DEPTH, torch.ones_like(Z), Z)
    iz = 1.0 / Zc
    Jp = torch.stack([fx * iz, o, -fx * X * iz * iz, o,
                      o, fy * iz, -fy * Y * iz * iz, o], dim=-1).view(1, N, H, W, 2, 4)
    Jj = torch.matmul(Jp, Ja)
    Ji = -SE3(Gij.data[:, :, None, None, None]).adjT(Jj)
    Jz = torch.matmul(Jp, G4.act(Jz0).unsqueeze(-1))
    return coords, valid, (Ji, Jj, Jz)
