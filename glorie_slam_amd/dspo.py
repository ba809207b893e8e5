"""DSPO stage 2 ("depth_scale"): host side of /root/reference/src/depth_video.py:222-285.

Order of operations kept from the reference:
  1. update_valid_depth_mask(up=False)           multiview consistency on the 1/8 maps
  2. align_scale_and_shift(mono, est, valid)     initial per-frame scale/shift
  3. mono_thres edge filtering                   -> an `edge_on` byte mask (no tensor compaction)
  4. `itrs` x BA_with_scale_shift                -> one glorie_dspo_scale_shift call
Differences: state is updated in place instead of rebinding attributes; the only
host synchronisation is the scalar "are there any edges left" needed for the fallback decision
of DepthVideo.ba (depth_video.py:290-294).
"""
import torch

from . import _lib as L
from .common import align_scale_and_shift


def scale_shift_step(video, target, weight, eta, ii, jj, edge_on, itrs, lm, ep, alpha=0.01):
    """glorie_dspo_scale_shift on the video buffers. target/weight [1,N,h,w,2] or [N,h,w,2]."""
    h, w = video.ht // video.down_scale, video.wd // video.down_scale
    target = target.reshape(-1, h, w, 2).contiguous().float()
    weight = weight.reshape(-1, h, w, 2).contiguous().float()
    eta = eta.reshape(-1, h, w).contiguous().float()
    N, M, B = ii.shape[0], eta.shape[0], video.disps.shape[0]
    vm = video.valid_depth_mask_small.contiguous().view(torch.uint8)      # bool storage is one byte per element
    eo = edge_on.to(torch.uint8).contiguous() if edge_on is not None else None
    L.check(L.load().glorie_dspo_scale_shift(
        video.ctx().handle, L.ptr(video.poses), L.ptr(video.disps), L.ptr(video.intrinsics),
        L.ptr(video.mono_disps), L.ptr(video.depth_scale), L.ptr(video.depth_shift), L.ptr(vm),
        L.ptr(target), L.ptr(weight), L.ptr(eta), L.ptr(ii.contiguous()), L.ptr(jj.contiguous()),
        L.ptr(eo), B, N, M, h, w, int(itrs), float(lm), float(ep), float(alpha), None,
        L.stream_ptr()), "glorie_dspo_scale_shift")


def _prepare_torch(video, n, ii, jj):
    """the reference's formulation, op by op (CPU tensors / cross-check for the fused kernels)"""
    video.update_valid_depth_mask(up=False)
    mono_d, est_d, valid_d = video.mono_disps[:n], video.disps[:n], video.valid_depth_mask_small[:n]
    scale_t, shift_t, error_t = align_scale_and_shift(mono_d, est_d, valid_d)
    video.depth_scale[:n] = scale_t
    video.depth_shift[:n] = shift_t
    if not video.mono_thres:
        return None, None
    avg = est_d.mean(dim=[1, 2])
    bad = (error_t / avg > video.mono_thres) | error_t.isnan() | (scale_t < 0) | \
          (valid_d.sum(dim=[1, 2]) < valid_d.shape[1] * valid_d.shape[2] * 0.5)
    bad_full = torch.zeros(video.disps.shape[0], dtype=torch.bool, device=bad.device)
    bad_full[:n] = bad
    edge_on = ~(bad_full[ii] | bad_full[jj])
    return edge_on, edge_on.any().to(torch.int32).reshape(1)


def depth_scale_stage(video, target, weight, eta, ii, jj, itrs, lm, ep, fused=True):
    """returns `success` like DepthVideo.dspo(opt_type='depth_scale') - or, for an eagerly issued step of an unsharded video on
    the GPU, the stage's device word `any edge left` (int32 [1]): the stage has then been enqueued in full (with every edge off it leaves
    all frames untouched) and the caller gates the stage-1 fallback on that word ON THE DEVICE (DepthVideo.ba,
    glorie_ba_set_gate) - no host decision, no stream drain per step"""
    n = video.counter.value
    self_publish = False
    if fused and video.disps.is_cuda and n > 0:
        from . import droid_backends
        mv = video.cfg['tracking']['multiview_filter']
        shard_ = getattr(video, "shard", None)
        # under hipGraph capture the stage-1 decision is left to the owner of the graph: the preparation's last launch
        # publishes the flag itself (a sharded run all-reduces it first and publishes below)
        self_publish = bool(video.mono_thres and torch.cuda.is_current_stream_capturing()
                            and not video.is_sharded())
        edge_on, any_on = droid_backends.dspo_prepare(
            video.poses, video.disps, video.intrinsics[0].contiguous(), video.mono_disps, n, mv['thresh'],
            mv['visible_num'], video.mono_thres, ii.contiguous(), jj.contiguous(), video.valid_depth_mask_small,
            video.depth_scale, video.depth_shift, publish=video.publish_target() if self_publish else None)
        if not video.mono_thres:
            edge_on = None
    else:
        edge_on, any_on = _prepare_torch(video, n, ii, jj)
    if edge_on is not None:
        if video.is_sharded():
            # the fallback decision must be identical on every rank (it selects a collective path); with the context's own
            # communicator the flag's all-reduce is stream work and the stage stays capturable
            from . import dist as gdist
            gdist.allreduce_flag_any(any_on, video.shard["group"], ctx=video._ctx, force=video.shard["force"])
        if any_on.is_cuda and not video.is_sharded() and not torch.cuda.is_current_stream_capturing():
            # EAGER steps (the tracking loop, whose edge set changes every keyframe): the decision stays on the device -
            # stage 2 is enqueued unconditionally (a no-op with every edge off) and the caller enqueues the stage-1 fallback
            # behind it, gated on this word.  No `.item()`, i.e. no stream drain per depth_scale stage: kept keyframe
            # 28.4 -> 27.5 ms (tools/prof_sequence.py, interleaved).  A RECORDED step keeps the pinned-word poll below: its
            # host wait overlaps the tail of the replay, while the ~12 gated no-op launches would cost the replay 3 %
            # (1068 -> 1040 it/s, tools/ab_bench.sh).
            if n <= 0:
                return False
            scale_shift_step(video, target, weight, eta, ii, jj, edge_on, itrs, lm, ep, alpha=0.01)
            video.disps.clamp_(min=1e-5)
            return any_on
        if any_on.is_cuda and torch.cuda.is_current_stream_capturing():
            # hipGraph capture: no host decision is possible here.  With every edge off the stage-2
            # launch below leaves all frames untouched, so it is recorded unconditionally; the flag
            # goes to pinned memory (tagged with a launch count, DepthVideo.publish_any_on) and the owner
            # of the graph (FactorGraph.update) runs the stage-1 fallback after the replay if it reads 0.
            if not self_publish:
                video.publish_any_on(any_on)
            video.deferred_fallback = True
        elif not bool(any_on.item()):   # the one scalar sync: decides the stage-1 fallback
            return False
    if n <= 0:
        return False
    scale_shift_step(video, target, weight, eta, ii, jj, edge_on, itrs, lm, ep, alpha=0.01)
    video.disps.clamp_(min=1e-5)
    return True
