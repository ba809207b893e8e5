"""Multi-GPU execution of the hot path: one process per GPU, `torch.distributed` over RCCL
(backend "nccl" on ROCm) -- SURVEY.md section 8(e).  The reference is single-GPU; this design is new.

* Bundle adjustment: edges are partitioned by SOURCE keyframe (`shard_frames`).  Per GN iteration
  the only exchange is ONE all-reduce(sum) of the packed fp64 system [H | v] (6P*6P + 6P doubles:
  14 KB at P = 7, 26 MB at P = 300); every rank then solves the same system and retracts its
  replica of the poses identically, and back-substitutes dz for the depth frames it owns.
  After the last iteration the owned disparity maps are all-gathered.
* Rendering: the cloud, its cell list and the decoder weights are replicated; rays are split in
  contiguous blocks (`shard_range`), no collective in the forward pass.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L


def shard_range(n, rank, world):
    """contiguous block [lo, hi) of n items for `rank`"""
    return rank * n // world, (rank + 1) * n // world


def shard_frames(ii_host, world):
    """Assign source keyframes to ranks in contiguous blocks with balanced edge counts.
    Returns owner[frame] (int array over 0..max frame).  Deterministic, identical on all ranks."""
    ii_host = np.asarray(ii_host, np.int64)
    nf = int(ii_host.max()) + 1 if ii_host.size else 0
    cnt = np.bincount(ii_host, minlength=nf).astype(np.float64)
    csum = np.cumsum(cnt)
    total = csum[-1] if nf else 0.0
    owner = np.zeros(nf, np.int64)
    for f in range(nf):
        mid = csum[f] - 0.5 * cnt[f]          # centre of mass of this frame's edges
        owner[f] = min(world - 1, int(mid * world / max(total, 1.0)))
    return owner


def local_edges(ii_host, owner, rank):
    """boolean mask of the edges whose source frame is owned by `rank`"""
    ii_host = np.asarray(ii_host, np.int64)
    return owner[ii_host] == rank


PACK_MIN_N6 = 96     # below this the exchange is latency-bound (6P = 42: 14 KB) and two extra launches cost more


def _active(group=None):
    return dist.is_available() and dist.is_initialized()


def init_ctx_comm(ctx, group=None, rank=None, world=None):
    """give the context its own RCCL communicator (glorie_comm_init): rank 0 draws the id, torch.distributed's object
    broadcast carries it (any backend), every rank joins.  With it the BA's exchange step is a C-ABI call on the stream
    (glorie_allreduce_normal_eq) instead of a torch.distributed collective between two ctypes calls - and the sharded
    iteration can be recorded into a hipGraph.  Without an initialised process group (one process) it forms a world of one."""
    lib = L.load()
    if _active(group):
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
    else:
        rank, world = 0, 1
    box = [None]
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        L.check(lib.glorie_comm_unique_id(buf), "glorie_comm_unique_id")
        box[0] = bytes(buf.raw)
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    L.check(lib.glorie_comm_init(ctx.handle, box[0], int(rank), int(world)), "glorie_comm_init")
    ctx._comm_group = (group, int(rank), int(world))        # allreduce_system checks its `group` against this
    return world


def ctx_comm_world(ctx):
    """ranks of the context's own communicator (0: none, the exchange goes through torch.distributed)"""
    return int(L.load().glorie_comm_world(ctx.handle)) if ctx is not None else 0


def native_comm(ctx, group=None, want_world=None):
    """True when `ctx` owns an RCCL communicator built for exactly the ranks of `group` (init_ctx_comm): the exchange steps
    are then C-ABI calls on the stream (capturable).  A caller that names another group - or a process group that appeared
    after a world-of-one communicator was made - goes through torch.distributed instead."""
    if ctx is None or ctx_comm_world(ctx) <= 0:
        return False
    made = getattr(ctx, "_comm_group", None)
    if want_world is None:
        want_world = dist.get_world_size(group) if _active(group) else 1
    return not (made is None or made[2] != want_world or (made[0] is not group and not (made[0] is None and group is None)))


def allreduce_flag_any(flag, group=None, ctx=None, force=False):
    """`flag` (int32 [1] on the device, 0 / 1) becomes 1 on every rank if it is 1 on any (the stage-1 fallback decision of
    a depth_scale stage selects a collective path: it has to be identical everywhere).  Native communicator: the flag
    rides as one double through glorie_allreduce_normal_eq - stream work, recorded into the step's hipGraph like the
    launches around it; otherwise a torch.distributed MAX."""
    if flag.is_cuda and native_comm(ctx, group):
        one = flag.to(torch.float64)
        L.check(L.load().glorie_allreduce_normal_eq(ctx.handle, L.ptr(one), 1, L.stream_ptr()), "glorie_allreduce_normal_eq")
        flag.copy_(one > 0)
        return flag
    if _active(group) and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return flag


def allreduce_system(hv, group=None, n6=None, force=False, ctx=None):
    """sum the reduced system [H | v] over ranks (RCCL all-reduce; gloo in the CPU tests).  hv: n6*n6 + n6
    doubles.  On the device and for n6 >= PACK_MIN_N6 only the lower triangle + v travel
    (glorie_ba_pack_system): xGMI rings are per-link bound, half the bytes is half the time.
    ctx with its own communicator (init_ctx_comm): the sum is glorie_allreduce_normal_eq on the current stream.
    force: run the collective even with one rank (exercises the RCCL path on a single GPU)."""
    native = hv.is_cuda and native_comm(ctx, group)
    if not native and (not _active(group) or (dist.get_world_size(group) <= 1 and not force)):
        return hv
    lib = L.load()

    def reduce(buf):
        if native:
            L.check(lib.glorie_allreduce_normal_eq(ctx.handle, L.ptr(buf), int(buf.numel()), L.stream_ptr()),
                    "glorie_allreduce_normal_eq")
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)

    if hv.is_cuda and n6 is not None and n6 >= PACK_MIN_N6:
        packed = torch.empty(n6 * (n6 + 1) // 2 + n6, dtype=torch.float64, device=hv.device)
        L.check(lib.glorie_ba_pack_system(L.ptr(hv), L.ptr(packed), int(n6), 0, L.stream_ptr()), "glorie_ba_pack_system")
        reduce(packed)
        L.check(lib.glorie_ba_pack_system(L.ptr(hv), L.ptr(packed), int(n6), 1, L.stream_ptr()), "glorie_ba_pack_system")
        return hv
    reduce(hv)
    return hv


def ba_sharded(ctx, poses, disps, intrinsics, targets, weights, eta, ii, jj, t0, t1, iterations,
               lm, ep, motion_only=False, depth_only=False, group=None, targets_hwc=False,
               force_collective=False):
    """Distributed droid_backends.ba: arguments are this rank's LOCAL edges (targets/weights
    [N,2,h,w], or [N,h,w,2] with targets_hwc; eta rows of unique(cat(arange(t0,t1), ii_local))); poses/disps are full replicas,
    updated in place (poses everywhere, disps for locally owned source frames)."""
    B, h, w = disps.shape
    N = int(ii.shape[0])
    P = t1 - t0
    n6 = 6 * P
    M = int(eta.shape[0]) if eta is not None else 0
    hv = torch.empty(n6 * n6 + n6, dtype=torch.float64, device=poses.device)
    dx = torch.zeros(P, 6, dtype=torch.float32, device=poses.device)
    lib = L.load()
    for _ in range(iterations):
        L.check(lib.glorie_ba_build_system(ctx.handle, L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), None,
                                           L.ptr(targets), L.ptr(weights), L.ptr(eta), L.ptr(ii), L.ptr(jj),
                                           B, N, M, h, w, int(t0), int(t1),
                                           int(bool(motion_only)) | (L.BA_TARGETS_HWC if targets_hwc else 0),
                                           L.ptr(hv), L.stream_ptr()), "glorie_ba_build_system")
        allreduce_system(hv, group, n6=n6, force=force_collective, ctx=ctx)
        L.check(lib.glorie_ba_solve_update(ctx.handle, L.ptr(poses), L.ptr(disps), L.ptr(ii), L.ptr(jj),
                                           B, N, M, h, w, int(t0), int(t1), float(lm), float(ep),
                                           int(bool(motion_only)), int(bool(depth_only)), L.ptr(hv),
                                           L.ptr(dx), None, L.stream_ptr()), "glorie_ba_solve_update")
    return dx


def _owner_blocks(owner, nf, world):
    """(start, count) per rank if every rank owns one contiguous block of the first nf frames, else None"""
    own = np.asarray(owner[:nf])
    if nf == 0 or np.any(np.diff(own) < 0):
        return None
    starts = np.searchsorted(own, np.arange(world), side="left")
    counts = np.searchsorted(own, np.arange(world), side="right") - starts
    return starts, counts


def allgather_owned_rows(buf, owner, rank, world, group=None, force=False, ctx=None):
    """make `buf[f]` consistent on all ranks: row f is taken from owner[f] (disps / depth_scale / disps_up
    after a sharded BA-update).  shard_frames hands every rank one contiguous block of frames, so this is an
    all-gather of the blocks (padded to the largest one): each row crosses a link once, where the masked
    all-reduce of full buffers it replaces moved every row twice and summed zeros.
    ctx with its own communicator for these ranks (init_ctx_comm): glorie_allgather_rows on the current stream."""
    if world <= 1 and not force:
        return buf
    native = buf.is_cuda and native_comm(ctx, group, want_world=world)
    nf = min(len(owner), buf.shape[0])
    blocks = _owner_blocks(owner, nf, world)
    if blocks is None:                      # non-contiguous ownership: masked all-reduce(sum)
        mine = torch.as_tensor(np.asarray(owner[:nf]) == rank, device=buf.device)
        contrib = torch.where(mine.view(-1, *([1] * (buf.dim() - 1))), buf[:nf], torch.zeros_like(buf[:nf]))
        if native:
            c64 = contrib.to(torch.float64).contiguous()
            L.check(L.load().glorie_allreduce_normal_eq(ctx.handle, L.ptr(c64), int(c64.numel()), L.stream_ptr()),
                    "glorie_allreduce_normal_eq")
            contrib = c64.to(buf.dtype)
        else:
            dist.all_reduce(contrib, op=dist.ReduceOp.SUM, group=group)
        buf[:nf] = contrib
        return buf
    starts, counts = blocks
    width = int(counts.max())
    send = torch.zeros((width,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    send[:counts[rank]] = buf[starts[rank]:starts[rank] + counts[rank]]
    recv = torch.empty((world * width,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    if native:
        L.check(L.load().glorie_allgather_rows(ctx.handle, L.ptr(send), L.ptr(recv), int(send.numel() * send.element_size()),
                                               L.stream_ptr()), "glorie_allgather_rows")
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    for r in range(world):
        if r != rank and counts[r]:
            buf[starts[r]:starts[r] + counts[r]] = recv[r * width:r * width + counts[r]]
    return buf
