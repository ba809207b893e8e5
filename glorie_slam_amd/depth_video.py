"""DepthVideo: shared keyframe state + the geometric operations of the hot path.

Host-side mirror of /root/reference/src/depth_video.py (same attribute and method names, same
argument meaning) whose native calls go through libglorie_hip:
  reproject -> glorie_reproject          (fused pops.projective_transform, no lietorch)
  distance  -> glorie_frame_distance
  ba / dspo -> glorie_ba (stage 1)  /  glorie_dspo_scale_shift (stage 2)
  upsample  -> glorie_cvx_upsample
  update_valid_depth_mask -> glorie_depth_filter

Deliberate difference, documented in DESIGN.md: DSPO stage 2 updates `disps`, `depth_scale`
and `depth_shift` IN PLACE; the reference rebinds the attributes to new tensors
(depth_video.py:278-282), which silently detaches the buffers shared with the mapper process.
"""
import os

import numpy as np
import torch
from torch.multiprocessing import Value

from . import droid_backends
from . import _lib as L


class DepthVideo:
    def __init__(self, cfg, printer=None):
        self.cfg = cfg
        self.output = f"{cfg['data']['output']}/{cfg['setting']}/{cfg['scene']}" if 'data' in cfg else ''
        ht = self.ht = cfg['cam']['H_out']
        wd = self.wd = cfg['cam']['W_out']
        self.counter = Value('i', 0)
        buffer = cfg['tracking']['buffer']
        self.BA_type = cfg['tracking']['backend']['BA_type']
        self.mono_thres = cfg['tracking']['mono_thres']
        self.device = cfg['device']
        self.down_scale = s = 8
        dev = self.device
        f32 = dict(device=dev, dtype=torch.float)

        self.timestamp = torch.zeros(buffer, **f32)
        store_images = cfg['tracking'].get('store_images', True)
        self.images = torch.zeros(buffer if store_images else 0, 3, ht, wd, device=dev, dtype=torch.uint8)
        self.dirty = torch.zeros(buffer, device=dev, dtype=torch.bool)
        self.npc_dirty = torch.zeros(buffer, device=dev, dtype=torch.bool)
        self.poses = torch.zeros(buffer, 7, **f32)
        self.disps = torch.ones(buffer, ht // s, wd // s, **f32)
        self.disps_up = torch.zeros(buffer, ht, wd, **f32)
        self.intrinsics = torch.zeros(buffer, 4, **f32)
        self.mono_disps = torch.zeros(buffer, ht // s, wd // s, **f32)
        self.depth_scale = torch.zeros(buffer, **f32)
        self.depth_shift = torch.zeros(buffer, **f32)
        self.valid_depth_mask = torch.zeros(buffer, ht, wd, device=dev, dtype=torch.bool)
        self.valid_depth_mask_small = torch.zeros(buffer, ht // s, wd // s, device=dev, dtype=torch.bool)
        self.fmaps = torch.zeros(buffer, 1, 128, ht // s, wd // s, dtype=torch.half, device=dev)
        self.nets = torch.zeros(buffer, 128, ht // s, wd // s, dtype=torch.half, device=dev)
        self.inps = torch.zeros(buffer, 128, ht // s, wd // s, dtype=torch.half, device=dev)
        self.poses[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1], **f32)
        self.printer = printer
        self._ctx = None
        # depth_scale stages that fell back to pose_depth (depth_video.py:290-294): decided on the host (CPU tensors, sharded
        # runs) or on the device (glorie_ba_set_gate; counted there, read lazily by the `stage2_fallbacks` property)
        self._fb_host = 0
        self._fb_hits = None
        # multi-GPU state (glorie_slam_amd.dist): None = single GPU
        self.shard = None

    # ---- bookkeeping -----------------------------------------------------------------
    @property
    def stage2_fallbacks(self):
        """host-decided fallbacks + the device's count of gated stage-1 BAs that ran (reading it synchronises)"""
        return self._fb_host + (int(self._fb_hits.item()) if self._fb_hits is not None else 0)

    @stage2_fallbacks.setter
    def stage2_fallbacks(self, value):
        self._fb_host = int(value) - (int(self._fb_hits.item()) if self._fb_hits is not None else 0)

    def count_host_fallback(self, n=1):
        """a stage-1 fallback decided on the host: no read of the device counter (the property drains the stream)"""
        self._fb_host += int(n)

    def get_lock(self):
        return self.counter.get_lock()

    def _set(self, index, item):
        if isinstance(index, int) and index >= self.counter.value:
            self.counter.value = index + 1
        elif isinstance(index, torch.Tensor) and index.max().item() > self.counter.value:
            self.counter.value = index.max().item() + 1
        self.timestamp[index] = item[0]
        if self.images.shape[0]:
            self.images[index] = item[1]
        if item[2] is not None:
            self.poses[index] = item[2]
        if item[3] is not None:
            self.disps[index] = item[3]
        if item[4] is not None:
            s = self.down_scale
            mono = item[4][s // 2 - 1::s, s // 2 - 1::s]
            self.mono_disps[index] = torch.where(mono > 0, 1.0 / mono, 0)
        if item[5] is not None:
            self.intrinsics[index] = item[5]
        if len(item) > 6:
            self.fmaps[index] = item[6]
        if len(item) > 7:
            self.nets[index] = item[7]
        if len(item) > 8:
            self.inps[index] = item[8]

    def __setitem__(self, index, item):
        with self.get_lock():
            self._set(index, item)

    def __getitem__(self, index):
        with self.get_lock():
            if isinstance(index, int) and index < 0:
                index = self.counter.value + index
            return (self.poses[index], self.disps[index], self.intrinsics[index],
                    self.fmaps[index], self.nets[index], self.inps[index])

    def append(self, *item):
        with self.get_lock():
            self._set(self.counter.value, item)

    @staticmethod
    def format_indicies(ii, jj, device="cuda"):
        if not isinstance(ii, torch.Tensor):
            ii = torch.as_tensor(ii)
        if not isinstance(jj, torch.Tensor):
            jj = torch.as_tensor(jj)
        ii = ii.to(device=device, dtype=torch.long).reshape(-1)
        jj = jj.to(device=device, dtype=torch.long).reshape(-1)
        return ii, jj

    def enable_sharding(self, owner, rank, world, group=None, force=False):
        """edges of this process' graphs are a source-keyframe shard (dist.shard_frames).
        On the GPU with RCCL as the process group's backend the context gets its OWN communicator (dist.init_ctx_comm), and
        every exchange of a BA-update - the all-reduce of the reduced normal equations, the fallback flag of a depth_scale
        stage, the owned rows - is a C-ABI call on the stream (glorie_allreduce_normal_eq / glorie_allgather_rows): the whole
        sharded step is then ONE hipGraph (FactorGraph.update).  GLORIE_NATIVE_COMM=0 opts out (torch.distributed
        collectives, the BA issued eagerly behind the replayed part); =1 asks for it whatever the backend says.  A
        communicator that cannot be created (or does not sum a probe correctly) leaves the torch.distributed path in place.
        force (tests): treat a world of one as sharded, so the exchange path runs on the single GPU of the test box."""
        self.shard = dict(owner=owner, rank=rank, world=world, group=group, force=bool(force))
        want = os.environ.get("GLORIE_NATIVE_COMM")
        if want == "0" or not self.poses.is_cuda or not (world > 1 or force):
            return
        import torch.distributed as tdist
        from . import dist as gdist
        active = tdist.is_available() and tdist.is_initialized()
        if want != "1" and world > 1 and not (active and tdist.get_backend(group) == "nccl"):
            return            # (gloo test runs with several ranks on one device: RCCL refuses duplicate devices)
        ok = True
        try:
            gdist.init_ctx_comm(self.ctx(), group, rank=rank if not active else None, world=world if not active else None)
            probe = torch.ones(1, dtype=torch.float64, device=self.poses.device)
            L.check(L.load().glorie_allreduce_normal_eq(self.ctx().handle, L.ptr(probe), 1, L.stream_ptr()),
                    "glorie_allreduce_normal_eq")
            if float(probe.item()) != float(world):
                raise RuntimeError(f"probe all-reduce summed to {float(probe.item())} over {world} ranks")
            # ... and the same collective RECORDED into a hipGraph and replayed (what FactorGraph.update will do with the
            # whole sharded step): a communicator that cannot be captured must not be discovered in the middle of a run
            probe.fill_(1.0)
            side = torch.cuda.Stream(self.poses.device)
            side.wait_stream(torch.cuda.current_stream(self.poses.device))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    L.check(L.load().glorie_allreduce_normal_eq(self.ctx().handle, L.ptr(probe), 1, L.stream_ptr()),
                            "glorie_allreduce_normal_eq")
                finally:
                    g.capture_end()
            g.replay()
            torch.cuda.synchronize(self.poses.device)
            if float(probe.item()) != float(world):
                raise RuntimeError(f"replayed probe all-reduce summed to {float(probe.item())} over {world} ranks")
            del g
        except Exception as exc:
            import warnings
            warnings.warn(f"context-owned RCCL communicator unavailable ({exc!r}): exchange through torch.distributed")
            ok = False
        if active and world > 1:
            # the ranks must AGREE on the exchange path: one rank falling back alone would meet the others' native
            # all-reduce with a torch.distributed collective and deadlock.  (A rank that failed before the id broadcast of
            # init_ctx_comm leaves its peers inside that broadcast - bounded by torch.distributed's own timeout.)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.poses.device)
            tdist.all_reduce(flag, op=tdist.ReduceOp.MIN, group=group)
            if ok and int(flag.item()) == 0:
                import warnings
                warnings.warn("another rank could not create its RCCL communicator: exchange through torch.distributed")
                ok = False
        if not ok:
            try:
                L.load().glorie_comm_destroy(self.ctx().handle)
            except Exception:
                pass

    def is_sharded(self):
        return self.shard is not None and (self.shard["world"] > 1 or self.shard.get("force", False))

    def native_exchange(self):
        """the exchange steps run on the context's own communicator (capturable)"""
        if not self.is_sharded():
            return False
        from . import dist as gdist
        return self.poses.is_cuda and gdist.native_comm(self._ctx, self.shard["group"], want_world=self.shard["world"])

    def sync_owned(self, *names):
        """all-gather the rows owned by each rank of the named per-keyframe buffers"""
        if not self.is_sharded():
            return
        from . import dist as gdist
        for nme in names:
            gdist.allgather_owned_rows(getattr(self, nme), self.shard["owner"], self.shard["rank"],
                                       self.shard["world"], self.shard["group"], force=self.shard["force"], ctx=self._ctx)
            if nme == "disps_up":
                self.shard["stale_up"] = False

    def sync_owned_state(self):
        """disps, depth_scale and depth_shift of every rank's frames in ONE collective (three small
        exchanges per BA-update are three collective latencies)"""
        if not self.is_sharded():
            return
        from . import dist as gdist
        nf = min(len(self.shard["owner"]), self.disps.shape[0])
        hw = self.disps.shape[1] * self.disps.shape[2]
        pack = torch.empty(nf, hw + 2, dtype=self.disps.dtype, device=self.disps.device)
        pack[:, :hw] = self.disps[:nf].reshape(nf, hw)
        pack[:, hw] = self.depth_scale[:nf]
        pack[:, hw + 1] = self.depth_shift[:nf]
        gdist.allgather_owned_rows(pack, self.shard["owner"], self.shard["rank"], self.shard["world"],
                                   self.shard["group"], force=self.shard["force"], ctx=self._ctx)
        self.disps[:nf] = pack[:, :hw].reshape(nf, *self.disps.shape[1:])
        self.depth_scale[:nf] = pack[:, hw]
        self.depth_shift[:nf] = pack[:, hw + 1]

    def mark_upsampled(self):
        """a sharded BA-update wrote the full-resolution disparities of this rank's frames only.  Nothing
        inside the update loop reads another rank's disps_up (61 MB at 50 keyframes), so their exchange is
        left to the consumers - the valid-depth mask, the mapper's accessors, save_video - via
        `fresh_disps_up()`; a COLLECTIVE call: every rank has to reach the same consumer."""
        if self.is_sharded():
            self.shard["stale_up"] = True

    def fresh_disps_up(self):
        if self.shard is not None and self.shard.get("stale_up"):
            self.sync_owned("disps_up")
        return self.disps_up

    def ctx(self):
        if self._ctx is None:
            self._ctx = L.Context()
        return self._ctx

    # ---- geometry --------------------------------------------------------------------
    def upsample(self, ix, mask, softmax_f32=False):
        """disps_up[ix] = cvx_upsample(disps[ix], mask)  (depth_video.py:140-144); mask may be the unevaluated logits of
        FusedUpdate (update_ops.LazyUpmask): convolution and upsampling are then one launch"""
        from .update_ops import LazyUpmask, conv_upsample
        if isinstance(mask, LazyUpmask):
            conv_upsample(mask, self.disps, ix.contiguous(), self.disps_up, softmax_f32=softmax_f32)
            return
        m = mask.reshape(-1, 576, mask.shape[-2], mask.shape[-1])
        if not m.is_contiguous() and not (m.dtype == torch.float16 and
                                          m.is_contiguous(memory_format=torch.channels_last)):
            m = m.contiguous()
        droid_backends.cvx_upsample(self.disps, ix.contiguous(), m, self.disps_up, softmax_f32=softmax_f32)

    def normalize(self):
        with self.get_lock():
            n = self.counter.value
            s = self.disps[:n].mean()
            self.disps[:n] /= s
            self.poses[:n, :3] *= s
            self.set_dirty(0, n)

    def reproject(self, ii, jj, motion=None):
        """coords [1,N,h,w,2], valid [1,N,h,w,1]  (depth_video.py:156-164); motion: see droid_backends.reproject"""
        ii, jj = DepthVideo.format_indicies(ii, jj, self.device)
        coords, valid = droid_backends.reproject(self.poses, self.disps, self.intrinsics, ii, jj, motion=motion)
        return coords[None], valid[None]

    def distance(self, ii=None, jj=None, beta=0.3, bidirectional=True):
        return_matrix = False
        if ii is None:
            return_matrix = True
            N = self.counter.value
            ii, jj = torch.meshgrid(torch.arange(N), torch.arange(N), indexing="ij")
        ii, jj = DepthVideo.format_indicies(ii, jj, self.device)
        intr0 = self.intrinsics[0].contiguous()
        if bidirectional:
            poses = self.poses[:self.counter.value].clone()
            d1 = droid_backends.frame_distance(poses, self.disps, intr0, ii, jj, beta)
            d2 = droid_backends.frame_distance(poses, self.disps, intr0, jj, ii, beta)
            d = .5 * (d1 + d2)
        else:
            d = droid_backends.frame_distance(self.poses, self.disps, intr0, ii, jj, beta)
        return d.reshape(N, N) if return_matrix else d

    # ---- bundle adjustment -----------------------------------------------------------
    def deferred_flag_init(self):
        """buffers of the deferred stage-1 decision (allocated outside any hipGraph capture): a pinned host word the device
        stores (launch count << 1 | any edge enabled) into, the device-side launch counter, the host's own count"""
        if getattr(self, "_flag_host", None) is None:
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._flag_np = self._flag_host.numpy()
            self._flag_count = torch.zeros(2, dtype=torch.int32, device=self.device)      # launch count, arrivals
            self._flag_expected = 0

    def publish_target(self):
        """(state, host word address) for droid_backends.dspo_prepare(publish=...): the preparation publishes itself"""
        self.deferred_flag_init()
        return self._flag_count, self._flag_host.data_ptr()

    def publish_any_on(self, any_on):
        """record the store of a depth_scale stage's `any edge enabled` into the pinned word (csrc/dspo_prep.hip)"""
        from . import _lib as L
        self.deferred_flag_init()
        L.check(L.load().glorie_publish_flag(L.ptr(any_on), L.ptr(self._flag_count), self._flag_host.data_ptr(),
                                             L.stream_ptr()), "glorie_publish_flag")

    def await_any_on(self, timeout=0.05):
        """the flag of the deferred stage replayed last.  Polls the pinned word for this replay's launch count: the host
        learns the decision when the preparation kernels are through (the rest of the replay is still running) and can
        enqueue the next step behind it; a stream synchronisation is only the fallback."""
        import time
        self._flag_expected += 1
        want = self._flag_expected
        t_end = time.perf_counter() + timeout
        while True:
            word = int(self._flag_np[0])
            if (word >> 1) == want:
                return word & 1
            if time.perf_counter() > t_end:
                torch.cuda.current_stream().synchronize()
                word = int(self._flag_np[0])
                if (word >> 1) > want:
                    # the device is AHEAD: an earlier replay's flag was never awaited (e.g. an exception between
                    # graph.replay() and this call).  After the stream synchronisation the word holds the decision of the
                    # last publish launch, which is this replay's - follow the device count from here on instead of
                    # failing every later step
                    import warnings
                    warnings.warn("deferred depth_scale decision: launch count %d, expected %d - resynchronised"
                                  % (word >> 1, want))
                    self._flag_expected = word >> 1
                elif (word >> 1) < want:
                    # the device is BEHIND: this replay contained no publish launch, the word is an older step's decision
                    self._flag_expected = word >> 1
                    raise RuntimeError("deferred depth_scale decision: the replay published nothing (launch count %d, "
                                       "expected %d)" % (word >> 1, want))
                return word & 1

    def dspo(self, target, weight, eta, ii, jj, t0=1, t1=None, itrs=2, lm=1e-4, ep=0.1,
             motion_only=False, opt_type="pose_depth", gate=None):
        with self.get_lock():
            if t1 is None:
                t1 = max(ii.max().item(), jj.max().item()) + 1
            h, w = self.ht // self.down_scale, self.wd // self.down_scale
            if opt_type == "pose_depth":
                # the kernels read the [N,h,w,2] layout the graph keeps them in: no permute + copy
                # (depth_video.py:215-216 of the reference)
                target = target.reshape(-1, h, w, 2).contiguous()
                weight = weight.reshape(-1, h, w, 2).contiguous()
                if self.is_sharded():
                    from . import dist as gdist
                    gdist.ba_sharded(self.ctx(), self.poses, self.disps, self.intrinsics[0].contiguous(),
                                     target, weight, eta.contiguous(), ii, jj, t0, t1, itrs, lm, ep,
                                     motion_only, False, group=self.shard["group"], targets_hwc=True,
                                     force_collective=self.shard["force"])
                else:
                    droid_backends.ba(self.poses, self.disps, self.intrinsics[0].contiguous(), None,
                                      target, weight, eta, ii, jj, t0, t1, itrs, lm, ep, motion_only,
                                      False, ctx=self.ctx(), want_updates=False, targets_hwc=True, gate=gate)
                self.disps.clamp_(min=1e-5)
                return True
            elif opt_type == "depth_scale":
                from . import dspo as _dspo
                return _dspo.depth_scale_stage(self, target, weight, eta, ii, jj, itrs, lm, ep)
            raise NotImplementedError(opt_type)

    def ba(self, target, weight, eta, ii, jj, t0=1, t1=None, iters=2, lm=1e-4, ep=0.1,
           motion_only=False, opt_type="pose_depth", eta_fallback=None):
        """eta_fallback: damping for the pose_depth slots, used when a depth_scale stage falls back to stage 1 and the
        two stages do not see the same depth frames (a shard's depth_scale stage covers the source frames of its own
        edges, its pose_depth BA the whole window: dist.ba_sharded)"""
        if self.BA_type == "DSPO":
            ok = self.dspo(target, weight, eta, ii, jj, t0, t1, iters, lm, ep, motion_only, opt_type)
            if isinstance(ok, torch.Tensor):
                # the stage left its decision on the device (dspo.depth_scale_stage): the reference's `if not success:
                # pose_depth` (depth_video.py:290-294) is enqueued behind it with every kernel gated on that word
                if self._fb_hits is None:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("the fallback counter must exist before a capture (run one eager step first)")
                    self._fb_hits = torch.zeros(1, dtype=torch.int32, device=ok.device)
                self.dspo(target, weight, eta if eta_fallback is None else eta_fallback, ii, jj, t0, t1, iters, lm, ep,
                          motion_only, "pose_depth", gate=(ok, self._fb_hits))
                return
            if not ok:
                self.count_host_fallback()
                self.dspo(target, weight, eta if eta_fallback is None else eta_fallback, ii, jj, t0, t1, iters, lm, ep,
                          motion_only, "pose_depth")
        elif self.BA_type == "DBA":
            self.dspo(target, weight, eta, ii, jj, t0, t1, iters, lm, ep, motion_only, "pose_depth")
        else:
            raise NotImplementedError(self.BA_type)

    # ---- masks -----------------------------------------------------------------------
    @torch.no_grad()
    def update_valid_depth_mask(self, up=True, fused=True):
        """two-view consistency mask (depth_video.py:326-361).  On the GPU the statistics, the depth
        filter, the exact nanmedian (radix select) and the threshold are 12 launches of
        glorie_valid_depth_mask; fused=False keeps the reference's op-by-op formulation."""
        if up:
            with self.get_lock():
                dirty_index, = torch.where(self.dirty.clone())
            if len(dirty_index) == 0:
                return
        else:
            dirty_index = torch.arange(self.counter.value, device=self.device)
        src = self.fresh_disps_up() if up else self.disps
        intr = (self.intrinsics[0].detach() * (self.down_scale if up else 1.0)).contiguous()
        mv = self.cfg['tracking']['multiview_filter']
        if fused and src.is_cuda:
            masks = droid_backends.valid_depth_mask(self.poses, src, intr, dirty_index.contiguous(), mv['thresh'],
                                                    mv['visible_num'])
        else:
            disps = torch.index_select(src, 0, dirty_index)
            depths = 1.0 / disps
            thresh = (mv['thresh'] * depths.mean(dim=[1, 2])).contiguous()
            count = droid_backends.depth_filter(self.poses, src, intr, dirty_index.contiguous(), thresh)
            depths[~(count >= mv['visible_num'])] = torch.nan
            med = depths.view(depths.shape[0], -1).nanmedian(dim=1).values
            masks = depths < 3 * med[:, None, None]
        if up:
            self.valid_depth_mask[dirty_index] = masks
            self.dirty[dirty_index] = False
        else:
            self.valid_depth_mask_small[dirty_index] = masks

    # ---- outputs ------------------------------------------------------
    def get_depth_scale_and_shift(self, index, mono_depth, est_depth, weights):
        """depth_video.py:301-311: weighted least-squares alignment of a mono depth map to the estimated depth
        of keyframe `index`; stores and returns [scale, shift]"""
        from .common import align_scale_and_shift
        scale, shift, _ = align_scale_and_shift(mono_depth, est_depth, weights)
        self.depth_scale[index] = scale
        self.depth_shift[index] = shift
        return [self.depth_scale[index], self.depth_shift[index]]

    def get_pose(self, index, device):
        """camera-to-world 4x4 matrix of keyframe `index` (depth_video.py:313-316: SE3(pose).inv().matrix());
        poses are stored world-to-camera as [tx ty tz qx qy qz qw]"""
        p = self.poses[index].detach().to(device=device, dtype=torch.float32)
        t, (qx, qy, qz, qw) = p[:3], p[3:]
        R = torch.stack([torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)]),
                         torch.stack([2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)]),
                         torch.stack([2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)])])
        c2w = torch.eye(4, dtype=torch.float32, device=p.device)
        c2w[:3, :3] = R.t()
        c2w[:3, 3] = -(R.t() @ t)
        return c2w

    def get_depth_and_pose(self, index, device):
        """(depth [H,W], valid mask [H,W], c2w [4,4]) of keyframe `index` (depth_video.py:318-324)"""
        with self.get_lock():
            est_depth = 1.0 / self.fresh_disps_up()[index].clone().to(device)
            depth_mask = self.valid_depth_mask[index].clone().to(device)
            c2w = self.get_pose(index, device)
        return est_depth, depth_mask, c2w

    def save_video(self, path):
        """video.npz of the reference (depth_video.py:367-384): poses [n,4,4] c2w, depths [n,H,W],
        timestamps [n], valid_depth_masks [n,H,W] (bool) for the n = counter keyframes"""
        poses, depths, stamps, masks = [], [], [], []
        for i in range(self.counter.value):
            depth, mask, pose = self.get_depth_and_pose(i, "cpu")
            poses.append(pose)
            depths.append(depth)
            stamps.append(self.timestamp[i].cpu())
            masks.append(mask)
        import numpy as np
        np.savez(path, poses=torch.stack(poses, 0).numpy(), depths=torch.stack(depths, 0).numpy(),
                 timestamps=torch.stack(stamps, 0).numpy(), valid_depth_masks=torch.stack(masks, 0).numpy())

    def set_dirty(self, index_start, index_end):
        self.dirty[index_start:index_end] = True
        self.npc_dirty[index_start:index_end] = True
