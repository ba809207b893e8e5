"""Mirror of /root/reference/src/geom/ba.py:127-216: `BA_with_scale_shift` with the reference's
arguments and return values, as ONE glorie_dspo_scale_shift launch (csrc/dspo.hip: Jz per pixel, per-frame 2x2 Schur
complement instead of the reference's dense M x M system + `schur_solve`).  DepthVideo.dspo calls the kernel on its
own buffers in place (glorie_slam_amd/dspo.py); this entry point is for callers that hold the reference's tensors."""
import torch

from . import _lib as L
from .lie import SE3


@torch.no_grad()
def BA_with_scale_shift(target, weight, eta, poses, disps, intrinsics, ii, jj, mono_disps, scales=None, shifts=None,
                        valid_depth_mask=None, ignore_frames=0, lm=0.0001, ep=0.1, alpha=1.0, fixedp=1, rig=1):
    """target / weight [1,N,h,w,2], eta [M,h,w] (M = unique(ii)), poses SE3 [1,B], disps / mono_disps / valid_depth_mask
    [1,B,h,w], intrinsics [1,B,4], scales / shifts [1,B] -> (poses, disps [1,B,h,w], wqs [1,B,2]); inputs untouched"""
    if ignore_frames != 0 or rig != 1:
        raise NotImplementedError("BA_with_scale_shift: the reference only ever passes ignore_frames=0, rig=1")
    data = poses.data if isinstance(poses, SE3) else poses
    L.need_cuda(data, disps, target, weight, ii, jj)
    _, B, h, w = disps.shape
    N = int(ii.shape[0])
    M = int(torch.unique(ii).numel())
    out = disps[0].clone().float().contiguous()
    sc = scales[0].clone().float().contiguous()
    sh = shifts[0].clone().float().contiguous()
    vm = valid_depth_mask[0].to(torch.bool).contiguous().view(torch.uint8)
    tg = target.reshape(N, h, w, 2).contiguous().float()
    wt = weight.reshape(N, h, w, 2).contiguous().float()
    et = eta.reshape(M, h, w).contiguous().float()
    ps = data[0].contiguous().float()
    intr = intrinsics[0].contiguous().float()
    mono = mono_disps[0].contiguous().float()
    ctx = L.default_context()
    L.check(L.load().glorie_dspo_scale_shift(
        ctx.handle, L.ptr(ps), L.ptr(out), L.ptr(intr), L.ptr(mono), L.ptr(sc), L.ptr(sh), L.ptr(vm), L.ptr(tg),
        L.ptr(wt), L.ptr(et), L.ptr(ii.long().contiguous()), L.ptr(jj.long().contiguous()), None, B, N, M, h, w, 1,
        float(lm), float(ep), float(alpha), None, L.stream_ptr()), "glorie_dspo_scale_shift")
    return poses, out[None], torch.stack([sc, sh], dim=-1)[None]
