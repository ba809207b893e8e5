"""FactorGraph: edge store + one DSPO BA-update iteration (`update`, `update_lowmem`).

Host-side mirror of /root/reference/src/factor_graph.py with the same public methods and
arguments.  What changed underneath:
  * edge lists are mirrored on the host, so `t0`, `unique(ii)` and the duplicate-edge filter
    need no `.item()` / `.cpu()` round trips per call (factor_graph.py:42-53,229-230);
  * reproject, the 4-level correlation lookup, BA and the convex upsampling are single HIP
    launches (see droid_backends / depth_video);
  * topology construction (`add_proximity_factors`, `add_backend_proximity_factors`) runs on
    numpy copies of the distance matrix instead of per-element tensor indexing -- same
    decisions, same edge order.
"""
import numpy as np
import os

import torch

from . import droid_backends
from .droid_net import ArenaLookup, CorrArena, CorrBlock, AltCorrBlock, FusedUpdate, FusedLookup, OtfCorrBlock


def coords_grid(ht, wd, device):
    y, x = torch.meshgrid(torch.arange(ht, device=device).float(),
                          torch.arange(wd, device=device).float(), indexing="ij")
    return torch.stack([x, y], dim=-1)


class FactorGraph:
    def __init__(self, video, update_op, device="cuda:0", corr_impl="volume", max_factors=-1,
                 use_graphs=False, capture_after=1):
        """use_graphs: replay `update()` as a hipGraph while the edge set is unchanged (a BA-update
        iteration is ~100 short launches; issuing them from Python costs as much as running them).
        capture_after: eager sightings of a call before it is recorded (1: the second sighting records; the tracking
        frontend, whose edge set changes with every keyframe, passes 6 - see update())"""
        self.video = video
        self.update_op = update_op
        self.device = device
        self.max_factors = max_factors
        self.corr_impl = corr_impl
        self.ht = ht = video.ht // video.down_scale
        self.wd = wd = video.wd // video.down_scale
        self.coords0 = coords_grid(ht, wd, device=device)
        long0 = lambda: torch.as_tensor([], dtype=torch.long, device=device)
        self.ii, self.jj, self.age = long0(), long0(), long0()
        self.corr, self.net, self.inp = None, None, None
        self.damping = 1e-6 * torch.ones_like(self.video.disps)
        zero_tw = lambda: torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        self.target, self.weight = zero_tw(), zero_tw()
        self.ii_inac, self.jj_inac = long0(), long0()
        self.ii_bad, self.jj_bad = long0(), long0()
        self.target_inac, self.weight_inac = zero_tw(), zero_tw()
        self._uniq_cache = None
        self.use_graphs = bool(use_graphs) and str(device).startswith("cuda")
        self.capture_after = max(int(capture_after), 1)   # eager sightings of a call before it is recorded (see update())
        self.stats = {"captures": 0, "replays": 0, "eager": 0}     # hipGraph bookkeeping of update() (pipeline / bench)
        self._topo = 0                      # bumped whenever the edge set changes
        self._graphs = {}                   # (topology, arguments) -> captured update
        self._lowmem_update = None
        # fp16 channels-last form of the update operator with fused element-wise stages
        # (droid_net.FusedUpdate, csrc/gru.hip)
        self.fast_update = FusedUpdate(update_op, inplace=self.use_graphs) if str(device).startswith("cuda") else None
        # gate term of the context features per source keyframe instead of per edge (an attribute: tests compare both)
        self.share_context = True

    def _use_arena(self):
        """the HIP pyramid builder + slot arena (widths that are multiples of 8, fp16 maps on the GPU)"""
        return (str(self.device).startswith("cuda") and self.wd % 8 == 0 and self.video.fmaps.dtype == torch.float16
                and getattr(self, "use_arena", True))

    def _otf_block(self):
        """volume-free correlation operator over the stored feature maps (corr_impl == 'otf'),
        rebuilt when the feature buffer was written (new keyframe)"""
        fm = self.video.fmaps
        key = (fm._version, self.video.counter.value)
        if getattr(self, "_otf_key", None) != key:
            num, rig, ch, ht, wd = fm.shape
            self._otf = OtfCorrBlock(fm.view(1, num * rig, ch, ht, wd))
            self._otf_key = key
            self._otf_rig = rig
        return self._otf

    # ---- host mirrors ----------------------------------------------------------------
    @staticmethod
    def _host(t):
        return t.detach().cpu().numpy().astype(np.int64) if t is not None and t.numel() else np.zeros(0, np.int64)

    def _unique_ii(self):
        """torch.unique(self.ii), cached until the edge set changes"""
        if self._uniq_cache is None:
            self._uniq_cache = (None, torch.unique(self.ii),
                                int(self.ii.min().item()) if self.ii.numel() else 0)
        return self._uniq_cache[1]

    def _groups(self):
        """(inverse index, #groups) of unique(ii) for GraphAgg, cached with the edge set"""
        self._unique_ii()
        if len(self._uniq_cache) < 4:
            uq, ix = torch.unique(self.ii, sorted=True, return_inverse=True)
            self._uniq_cache = self._uniq_cache + ((ix, int(uq.shape[0])),)
        return self._uniq_cache[3]

    def _filter_repeated_edges(self, ii, jj):
        have = set(zip(self._host(self.ii).tolist(), self._host(self.jj).tolist()))
        have |= set(zip(self._host(self.ii_inac).tolist(), self._host(self.jj_inac).tolist()))
        hi, hj = self._host(ii), self._host(jj)
        keep = torch.as_tensor([(a, b) not in have for a, b in zip(hi.tolist(), hj.tolist())],
                               dtype=torch.bool, device=ii.device)
        return ii[keep], jj[keep]

    def filter_edges(self):
        conf = torch.mean(self.weight, dim=[0, 2, 3, 4])
        mask = (torch.abs(self.ii - self.jj) > 2) & (conf < 0.001)
        self.ii_bad = torch.cat([self.ii_bad, self.ii[mask]])
        self.jj_bad = torch.cat([self.jj_bad, self.jj[mask]])
        self.rm_factors(mask, store=False)

    def clear_edges(self):
        for k in ("ii", "jj", "age", "corr", "damping", "net", "inp", "target", "weight", "ii_inac",
                  "jj_inac", "ii_bad", "jj_bad", "target_inac", "weight_inac"):
            setattr(self, k, None)

    # ---- edge management -------------------------------------------------------------
    @torch.no_grad()
    def add_factors(self, ii, jj, remove=False):
        if not isinstance(ii, torch.Tensor):
            ii = torch.as_tensor(ii, dtype=torch.long, device=self.device)
        if not isinstance(jj, torch.Tensor):
            jj = torch.as_tensor(jj, dtype=torch.long, device=self.device)
        ii, jj = self._filter_repeated_edges(ii, jj)
        if ii.shape[0] == 0:
            return
        if self.max_factors > 0 and self.ii.shape[0] + ii.shape[0] > self.max_factors \
                and self.corr is not None and remove:
            # equal ages keep their index order (what the reference's CUDA radix sort does; torch's CPU sort does not)
            ix = torch.arange(len(self.age))[torch.argsort(self.age, stable=True).cpu()]
            self.rm_factors(ix >= self.max_factors - ii.shape[0], store=True)
        net = self.video.nets[ii].to(self.device).unsqueeze(0)
        if self.corr_impl == "volume":
            c = (ii == jj).long()
            if self._use_arena():
                # slot-indexed store: the new edges are built into free slots (one launch), nothing else moves.  Only the
                # maps of the frames these edges touch are converted (channels-last, scaled by 1/4 as corr.py:70-71): the
                # buffer holds hundreds of frames and is written for every new keyframe
                num, rig, ch, fht, fwd = self.video.fmaps.shape
                n_new = int(ii.shape[0])
                frames, inv = torch.unique(torch.cat([rig * ii, rig * jj + c]), return_inverse=True)
                fm = self.video.fmaps.view(num * rig, ch, fht, fwd)[frames].to(self.device)
                fcl = (fm.half() / 4.0).permute(0, 2, 3, 1).reshape(int(frames.shape[0]), fht * fwd, ch).contiguous()
                if self.corr is None:
                    self.corr = CorrArena(self.ht, self.wd, self.device,
                                          capacity=max(16, self.max_factors + 8 if self.max_factors > 0 else n_new))
                self.corr.add(fcl, inv[:n_new].contiguous(), inv[n_new:].contiguous())
            else:
                fmap1 = self.video.fmaps[ii, 0].to(self.device).unsqueeze(0)
                fmap2 = self.video.fmaps[jj, c].to(self.device).unsqueeze(0)
                with torch.autocast("cuda", enabled=True):
                    corr = CorrBlock(fmap1, fmap2)
                self.corr = corr if self.corr is None else self.corr.cat(corr)
            inp = self.video.inps[ii].to(self.device).unsqueeze(0)
            self.inp = inp if self.inp is None else torch.cat([self.inp, inp], 1)
        elif self.corr_impl == "otf":
            inp = self.video.inps[ii].to(self.device).unsqueeze(0)
            self.inp = inp if self.inp is None else torch.cat([self.inp, inp], 1)
        target, _ = self.video.reproject(ii, jj)
        weight = torch.zeros_like(target)
        self._uniq_cache = None
        self._topo += 1
        self._graphs.clear()
        self.ii = torch.cat([self.ii, ii], 0)
        self.jj = torch.cat([self.jj, jj], 0)
        self.age = torch.cat([self.age, torch.zeros_like(ii)], 0)
        self.net = net if self.net is None else torch.cat([self.net, net], 1)
        self.target = torch.cat([self.target, target], 1)
        self.weight = torch.cat([self.weight, weight], 1)

    def rm_factors(self, mask, store=False):
        mask = mask.to(self.ii.device)
        self._uniq_cache = None
        self._topo += 1
        self._graphs.clear()
        if store:
            self.ii_inac = torch.cat([self.ii_inac, self.ii[mask]], 0)
            self.jj_inac = torch.cat([self.jj_inac, self.jj[mask]], 0)
            self.target_inac = torch.cat([self.target_inac, self.target[:, mask]], 1)
            self.weight_inac = torch.cat([self.weight_inac, self.weight[:, mask]], 1)
        keep = ~mask
        self.ii, self.jj, self.age = self.ii[keep], self.jj[keep], self.age[keep]
        if self.corr_impl == "volume" and self.corr is not None:
            if isinstance(self.corr, CorrArena):
                self.corr.keep(keep.cpu().tolist())
            else:
                self.corr = self.corr[keep]
        if self.net is not None:
            self.net = self.net[:, keep]
        if self.inp is not None:
            self.inp = self.inp[:, keep]
        self.target = self.target[:, keep]
        self.weight = self.weight[:, keep]

    def rm_keyframe(self, ix):
        v = self.video
        with v.get_lock():
            for name in ("timestamp", "images", "dirty", "npc_dirty", "poses", "disps", "disps_up",
                         "intrinsics", "depth_scale", "depth_shift", "mono_disps", "valid_depth_mask",
                         "valid_depth_mask_small", "nets", "inps", "fmaps"):
                buf = getattr(v, name)
                if buf.shape[0] > ix + 1:
                    buf[ix] = buf[ix + 1]
        m = (self.ii_inac == ix) | (self.jj_inac == ix)
        self.ii_inac[self.ii_inac >= ix] -= 1
        self.jj_inac[self.jj_inac >= ix] -= 1
        if torch.any(m):
            self.ii_inac, self.jj_inac = self.ii_inac[~m], self.jj_inac[~m]
            self.target_inac, self.weight_inac = self.target_inac[:, ~m], self.weight_inac[:, ~m]
        m = (self.ii == ix) | (self.jj == ix)
        self.ii[self.ii >= ix] -= 1
        self.jj[self.jj >= ix] -= 1
        self.rm_factors(m, store=False)

    # ---- one BA-update iteration -----------------------------------------------------
    def poses_on_device(self):
        return self.video.poses.is_cuda

    def _padded_flow(self, n, device):
        """the zero-padded fp16 motion map of the current edge set (update_ops.PaddedFlow), kept with the edge set"""
        from .update_ops import PaddedFlow
        pf = self._graphs.get("flow_pad")
        if pf is None or not pf.fits(n, self.ht, self.wd, device):
            pf = self._graphs["flow_pad"] = PaddedFlow(n, self.ht, self.wd, device)
        return pf

    def _motion(self, coords1, padded=False):
        """[1, N, 4, h, w] view of the channels-last motion map (factor_graph.py:219-221); padded=True: the zero-padded
        fp16 form FusedUpdate's flow encoder reads (update_ops.PaddedFlow), written directly"""
        if coords1.is_cuda and padded:
            n = coords1.numel() // (self.ht * self.wd * 2)
            pf = self._padded_flow(n, coords1.device)
            return droid_backends.motion_padded(coords1.contiguous(), self.coords0.contiguous(), self.target.contiguous(), pf)
        if coords1.is_cuda:
            m = droid_backends.motion(coords1.contiguous(), self.coords0.contiguous(), self.target.contiguous())
            return m.permute(0, 3, 1, 2).unsqueeze(0)
        motn = torch.cat([coords1 - self.coords0, self.target - coords1], dim=-1)
        return motn.permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)

    def _window(self, t0, t1, use_inactive):
        """(t0, t1) as the reference resolves them when the caller passes None (factor_graph.py:229-230:
        t0 = max(1, ii.min() + 1); depth_video.py:211-212: t1 = max over the BA's edge list + 1), from host
        mirrors of the edge lists cached with the edge set: no .item() per call, and a captured update
        contains no host decision"""
        if t0 is not None and t1 is not None:
            return t0, t1
        st = self._graphs.get("edge_stats")
        if st is None:
            hi, hj = self._host(self.ii), self._host(self.jj)
            st = self._graphs["edge_stats"] = (int(hi.min()), int(max(hi.max(), hj.max())),
                                               self._host(self.ii_inac), self._host(self.jj_inac))
        if t0 is None:
            t0 = max(1, st[0] + 1)
        if t1 is None:
            top = st[1]
            if use_inactive and st[2].size:
                m = (st[2] >= t0 - 3) & (st[3] >= t0 - 3)
                if m.any():
                    top = max(top, int(st[2][m].max()), int(st[3][m].max()))
            t1 = top + 1
        return t0, t1

    def _arena_generations(self):
        from . import _lib as L
        return (self.video.ctx().generation(), L.default_context().generation())

    @torch.no_grad()
    def update(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False,
               opt_type="pose_depth"):
        """factor_graph.py:212-256.  With use_graphs the launch sequence of one call is captured
        the second time it is seen for the current edge set and replayed afterwards."""
        sharded = self.video.is_sharded()
        # with the context's own RCCL communicator (DepthVideo.enable_sharding) every exchange of the step is stream work:
        # the WHOLE sharded step - update operator, build -> all-reduce -> solve per GN iteration, the all-reduced fallback
        # flag, upsampling, the exchange of the owned rows - is one hipGraph, like the single-GPU step.  Through
        # torch.distributed (gloo, or GLORIE_NATIVE_COMM=0) the replay stops before the BA, which is issued eagerly
        whole = not sharded or self.video.native_exchange()
        if self.ii.numel() == 0:
            return self._update_eager(t0, t1, itrs, use_inactive, EP, motion_only, opt_type)
        t0, t1 = self._window(t0, t1, use_inactive)
        if not self.use_graphs:
            return self._update_eager(t0, t1, itrs, use_inactive, EP, motion_only, opt_type)
        key = (self._topo, t0, t1, itrs, bool(use_inactive), float(EP), bool(motion_only), opt_type, sharded, whole)
        ent = self._graphs.get(key)
        if ent is None or isinstance(ent, int):
            # the first `capture_after` sightings of a call for this edge set run eagerly (the first one also packs weights
            # and sizes scratch buffers): a capture keeps the device idle for ~2.7 ms and a replay saves ~0.5 ms over an
            # eager call that is enqueued almost as fast as it executes - in the tracking loop, where every keyframe changes
            # the edge set and a call is seen ~5 times, capturing on the second sighting cost 10 % (tools/prof_sequence.py);
            # the BA-update loop of the bench and a backend that keeps its graph replay thousands of times
            n = (ent or 0) + 1
            self._graphs[key] = n if n < self.capture_after else "seen"
            return self._update_eager(t0, t1, itrs, use_inactive, EP, motion_only, opt_type)
        if ent == "seen":
            try:
                ent = self._capture(key, (t0, t1, itrs, use_inactive, EP, motion_only, opt_type, whole))
                self.stats["captures"] += 1
            except Exception as exc:        # a failed capture must not take the step down: stay eager
                import warnings
                warnings.warn(f"hipGraph capture of FactorGraph.update failed ({exc!r}); running eagerly")
                ent = self._graphs[key] = "eager"
        if ent == "eager":
            return self._update_eager(t0, t1, itrs, use_inactive, EP, motion_only, opt_type)
        if ent[-1] != self._arena_generations():
            # a scratch arena moved since the capture (another graph, the backend or a bigger KNN build grew
            # it): the recorded launches point into freed memory.  Drop them; this call runs eagerly and the
            # edge set is captured again on its next sighting.
            for k in [k for k, v in self._graphs.items() if isinstance(v, tuple) and k not in ("static", "edge_stats")
                      and isinstance(v[0], torch.cuda.CUDAGraph)]:
                self._graphs[k] = "seen"
            return self._update_eager(t0, t1, itrs, use_inactive, EP, motion_only, opt_type)
        graph, s_net, s_target, s_weight, ba_args, deferred, eta_fb, _gen = ent
        for dst, src in ((s_net, self.net), (s_target, self.target), (s_weight, self.weight)):
            if src is not dst:
                dst.copy_(src)
        graph.replay()
        self.stats["replays"] += 1
        self.net, self.target, self.weight = s_net, s_target, s_weight
        self._ba_args = ba_args
        self._eta_fb = eta_fb
        if not whole:
            return self._update_finish(itrs, motion_only, opt_type)
        if sharded:
            # host-side bookkeeping of the recorded step: the replay rewrote this rank's disps_up rows, the graph cannot
            # set the flag its consumers (valid-depth mask, mapper, save_video) look at
            self.video.mark_upsampled()
        if deferred:
            # the recorded depth_scale stage could not take its stage-1 fallback decision on the host
            # (dspo.depth_scale_stage): read the flag it leaves in pinned memory (no stream synchronisation:
            # DepthVideo.await_any_on polls for this replay's launch count) and redo it here.  Sharded: the flag was
            # all-reduced inside the replay, so every rank takes this (collective) branch together; the fallback BA of a
            # shard spans the whole window and has its own damping rows (eta_fb)
            if self.video.await_any_on() == 0:
                self.video.count_host_fallback()
                target, weight, damping, ii, jj, uniq, upmask, t0_, t1_ = ba_args
                self.video.dspo(target, weight, damping if eta_fb is None else eta_fb, ii, jj, t0_, t1_, itrs, 1e-4, 0.1,
                                motion_only, "pose_depth")
                self.video.upsample(uniq, upmask)
                if sharded:
                    self.video.sync_owned_state()
                    self.video.mark_upsampled()

    def _capture(self, key, args):
        """capture one update() on static copies of the recurrent state (net, target, weight);
        everything else it touches (video buffers, damping, age) is updated in place already"""
        # one set of static buffers per edge set, shared by every graph captured for it (the pose_depth and
        # depth_scale graphs alternate: private copies would cost three device copies per replay)
        st = self._graphs.get("static")
        if st is None:
            st = (self.net.clone(memory_format=torch.preserve_format), self.target.clone(), self.weight.clone())
            self._graphs["static"] = st
        s_net, s_target, s_weight = st
        for dst, src in ((s_net, self.net), (s_target, self.target), (s_weight, self.weight)):
            if src is not dst:
                dst.copy_(src)
        keep = (self.net, self.target, self.weight)
        self.video.deferred_flag_init()          # persistent buffers: not from the graph's private pool
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        self.net, self.target, self.weight = s_net, s_target, s_weight
        self.video.deferred_fallback = False
        try:
            # capture_begin / capture_end on a side stream instead of the `torch.cuda.graph` context manager: that one also
            # runs gc.collect() and torch.cuda.empty_cache() on entry - 3.5 ms of a 5.9 ms capture here, and every cached
            # block handed back to the driver has to be hipMalloc'ed again by the next eager step; the tracking loop records
            # ~2 graphs per keyframe and replays each ~4 times (tools/prof_sequence.py).
            # thread_local: calls of other threads (e.g. the RCCL watchdog of torch.distributed) must not
            # invalidate the capture
            # ONE memory pool for every graph of this object: the blocks of the graphs an edge-set change drops go back to it
            # and the next capture takes them.  With a private pool per graph every buffer of every capture is a hipMalloc
            # (0.5 ms each, ~35 per keyframe: the largest single item of the tracking loop's host profile) and the dropped
            # pools only return to the driver through empty_cache (100 GB reserved after 100 frames without it).  A
            # one-node keeper graph holds the pool open (torch asserts when a pool handle outlives its last graph).
            # Sharing is safe here: the graphs never run concurrently, and what a replay leaves behind for the host (ba_args)
            # is read before any other graph of the pool is replayed or captured.
            dev = self.net.device
            cap = getattr(self, "_capture_stream", None)
            if cap is None:
                cap = self._capture_stream = torch.cuda.Stream(dev)
                self._graph_pool = torch.cuda.graph_pool_handle()
                self._pool_keeper = torch.cuda.CUDAGraph()
                with torch.cuda.stream(cap):
                    self._pool_keeper.capture_begin(pool=self._graph_pool, capture_error_mode="thread_local")
                    self._pool_keeper_buf = torch.zeros(1, device=dev)
                    self._pool_keeper.capture_end()
            if torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev) > getattr(self, "capture_trim_bytes", 32 << 30):
                torch.cuda.empty_cache()                     # (other objects' dropped pools, eager-step leftovers)
            with torch.cuda.stream(cap):
                graph.capture_begin(pool=self._graph_pool, capture_error_mode="thread_local")
                try:
                    self._update_eager(*args)
                    if self.net.data_ptr() != s_net.data_ptr():     # FusedUpdate(inplace) already wrote s_net
                        s_net.copy_(self.net)
                    # the BA arguments alias the recurrent state: keep them pointing at the static copies
                    ba_args = tuple(s_target if a is self.target else (s_weight if a is self.weight else a)
                                    for a in self._ba_args)
                    if self.target is not s_target:                 # (the eager step adds in place when it can)
                        s_target.copy_(self.target)
                    s_weight.copy_(self.weight)
                finally:
                    graph.capture_end()
        finally:
            # the capture did not execute anything: restore the state the caller had
            self.net, self.target, self.weight = keep
        ent = (graph, s_net, s_target, s_weight, ba_args, bool(self.video.deferred_fallback),
               getattr(self, "_eta_fb", None), self._arena_generations())
        self._graphs[key] = ent
        return ent

    @torch.no_grad()
    def _update_eager(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False,
                      opt_type="pose_depth", run_ba=True):
        if run_ba and not torch.cuda.is_current_stream_capturing():
            self.stats["eager"] += 1
        if self.fast_update is not None and self.poses_on_device() and self.target.is_contiguous() \
                and self.target.dtype == torch.float32:
            # reprojection + motion features of the flow encoder (padded fp16 map) in one launch
            pf = self._padded_flow(int(self.ii.shape[0]), self.target.device)
            coords1, mask = self.video.reproject(self.ii, self.jj, motion=(self.target, pf))
            motn = pf
        else:
            coords1, mask = self.video.reproject(self.ii, self.jj)
            motn = self._motion(coords1, padded=self.fast_update is not None)
        if self.corr_impl == "otf":
            blk = self._otf_block()
            rig = self._otf_rig
            ck = ("otf_idx", rig)
            if ck not in self._graphs:
                self._graphs[ck] = ((rig * self.ii).contiguous(), (rig * self.jj + (self.ii == self.jj).long()).contiguous())
            oi, oj = self._graphs[ck]
            lookup = FusedLookup(blk, coords1, oi, oj) if (self.fast_update is not None and blk.num_levels == 4) \
                else (lambda: blk(coords1, oi, oj))
        elif self.fast_update is not None and isinstance(self.corr, CorrArena) and self.corr.layout == "dm":
            lookup = ArenaLookup(self.corr, coords1)      # lookup + corr_encoder[0] in one launch (csrc/corr_dm.hip)
        elif self.fast_update is not None and isinstance(self.corr, (CorrArena, CorrBlock)):
            lookup = lambda: self.corr(coords1, channels_last=True)      # what FusedUpdate's 1x1 encoder consumes
        else:
            lookup = lambda: self.corr(coords1)
        uniq = self._unique_ii()
        if self.fast_update is not None:
            # the lookup is handed over as a callable: FusedUpdate issues it behind the fork of its
            # independent branches (flow encoder, global context), which then overlap with it
            # the edges of one source keyframe share their context features (inp = video.inps[ii], add_factors):
            # the hoisted gate term is kept per keyframe and shared through the GraphAgg grouping
            ix, _ = self._groups()
            self.net, delta, weight, damping, upmask = \
                self.fast_update(self.net, self.inp, lookup, motn, self.ii, self.jj, self._groups(),
                                 context=(self.video.inps, uniq, ix) if self.share_context else None,
                                 lazy_up=True, weight_out=self.weight)
        else:
            corr = lookup()
            with torch.autocast("cuda", enabled=True):
                self.net, delta, weight, damping, upmask = \
                    self.update_op(self.net, self.inp, corr, motn, self.ii, self.jj)
        if t0 is None or t1 is None:
            t0, t1 = self._window(t0, t1, use_inactive)
        sharded = self.video.is_sharded()
        # target = coords1 + delta, damping[uniq] = eta, the BA's damping 0.2 * eta + EP and age += 1 in ONE launch
        # (glorie_update_bookkeeping) when the tensors are the plain fp32 ones of the fused operator
        fused_book = (self.fast_update is not None and not use_inactive and not sharded and coords1.is_cuda
                      and self.target.shape == coords1.shape and self.target.dtype == torch.float32
                      and self.target.is_contiguous() and coords1.is_contiguous() and coords1.dtype == torch.float32
                      and delta.dtype == torch.float32 and delta.is_contiguous() and delta.shape == coords1.shape
                      and damping.dtype == torch.float32 and damping.is_contiguous()
                      and self.damping.dtype == torch.float32 and self.damping.is_contiguous()
                      and damping.numel() == uniq.shape[0] * self.ht * self.wd and self.age.is_contiguous())
        damping_ba = None
        if fused_book:
            from . import _lib as L
            damping_ba = torch.empty((uniq.shape[0], self.ht, self.wd), dtype=torch.float32, device=coords1.device)
            L.check(L.load().glorie_update_bookkeeping(
                L.ptr(coords1), L.ptr(delta), L.ptr(self.target), coords1.numel(), L.ptr(damping), L.ptr(uniq),
                L.ptr(self.damping), L.ptr(damping_ba), int(uniq.shape[0]), self.ht * self.wd, float(EP), L.ptr(self.age),
                int(self.age.shape[0]), L.stream_ptr()), "glorie_update_bookkeeping")
            self._age_done = True
        elif self.target.shape == coords1.shape and self.target.dtype == torch.float32 and self.target.is_contiguous():
            torch.add(coords1, delta.to(dtype=torch.float), out=self.target)   # no new tensor, no copy when captured
        else:
            self.target = coords1 + delta.to(dtype=torch.float)
        self.weight = weight.to(dtype=torch.float)
        if not fused_book:
            self.damping[uniq] = damping.to(self.damping.dtype)
        if use_inactive:
            # the inactive factors that still touch the window (factor_graph.py:232-238) do not change while
            # the edge set stays the same: their indices, the frame list and one [inactive | active] buffer
            # per tensor are kept, every iteration only refreshes the active tail (the reference gathers
            # and concatenates ~4x the active data per iteration)
            ck = ("inac", t0)
            c = self._graphs.get(ck)
            if c is None:
                m = (self.ii_inac >= t0 - 3) & (self.jj_inac >= t0 - 3)
                ii_c = torch.cat([self.ii_inac[m], self.ii], 0)
                jj_c = torch.cat([self.jj_inac[m], self.jj], 0)
                tgt = torch.cat([self.target_inac[:, m], self.target], 1)
                wgt = torch.cat([self.weight_inac[:, m], self.weight], 1)
                c = (ii_c, jj_c, torch.unique(ii_c), tgt, wgt, tgt.shape[1] - self.target.shape[1])
                self._graphs[ck] = c
            ii, jj, uq, target, weight, n_inac = c
            target[:, n_inac:] = self.target
            weight[:, n_inac:] = self.weight
        else:
            ii, jj, target, weight, uq = self.ii, self.jj, self.target, self.weight, uniq
        if sharded and opt_type == "pose_depth":
            # a shard sees only its own source frames: the BA slots are unique(cat(arange(t0,t1), ii)),
            # cached per edge set (torch.unique synchronises with the host)
            assert t1 is not None, "sharded BA needs an explicit (global) t1"
            ck = ("uq", self._topo, t0, t1)
            if ck not in self._graphs:
                self._graphs[ck] = torch.unique(torch.cat([torch.arange(t0, t1, device=ii.device), ii]))
            uq = self._graphs[ck]
        if uq is uniq and damping_ba is not None:
            damping = damping_ba
        elif uq is uniq:
            # same frames as just written: 0.2 * eta + EP straight from the operator's output (no gather)
            damping = damping.reshape(-1, self.ht, self.wd).to(self.damping.dtype).mul(0.2).add_(EP)
        else:
            damping = .2 * self.damping[uq].contiguous() + EP
        if opt_type == "pose_depth" and not motion_only and not sharded:
            # ba_cuda views eta as [len(unique(cat(arange(t0, t1), ii))), h*w] and raises when the sizes
            # disagree (droid_kernels.cu:1339-1352); the device only sets a status bit, so the count is
            # checked here against the host mirror of the edge list (cached with the edge set)
            ck = ("slots", t0, t1, bool(use_inactive))
            if ck not in self._graphs:
                self._graphs[ck] = int(np.unique(np.concatenate([np.arange(t0, t1), self._host(ii)])).size)
            if damping.shape[0] != self._graphs[ck]:
                raise RuntimeError(f"FactorGraph.update: eta has {damping.shape[0]} frames, the BA window "
                                   f"[{t0}, {t1}) + the source frames of the edges span {self._graphs[ck]}")
        self._ba_args = (target, weight, damping, ii, jj, uniq, upmask, t0, t1)
        self._eta_fb = None
        if sharded and opt_type == "depth_scale":
            # a stage-1 fallback of this stage is a sharded pose_depth BA: its slots are the whole window + the source
            # frames of the local edges, not the frames `damping` was gathered for
            assert t1 is not None, "sharded BA needs an explicit (global) t1"
            ck = ("uq", self._topo, t0, t1)
            if ck not in self._graphs:
                self._graphs[ck] = torch.unique(torch.cat([torch.arange(t0, t1, device=ii.device), ii]))
            self._eta_fb = .2 * self.damping[self._graphs[ck]].contiguous() + EP
        if run_ba:
            self._update_finish(itrs, motion_only, opt_type)

    def _update_finish(self, itrs, motion_only, opt_type):
        """BA + upsampling (+ exchange of the owned rows when sharded) on the outputs of the update operator"""
        target, weight, damping, ii, jj, uniq, upmask, t0, t1 = self._ba_args
        self.video.ba(target, weight, damping, ii, jj, t0, t1, iters=itrs, lm=1e-4, ep=0.1,
                      motion_only=motion_only, opt_type=opt_type, eta_fallback=getattr(self, "_eta_fb", None))
        self.video.upsample(uniq, upmask)
        if self.video.is_sharded():
            self.video.sync_owned_state()
            self.video.mark_upsampled()
        if getattr(self, "_age_done", False):
            self._age_done = False                  # incremented by the bookkeeping launch of this update
        else:
            self.age += 1

    @torch.no_grad()
    def update_lowmem(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, steps=8,
                      enable_wq=True):
        """factor_graph.py:259-309 -- alt-corr, 8 source frames per chunk"""
        num, rig, ch, ht, wd = self.video.fmaps.shape
        # on-the-fly correlation: the MFMA operator (csrc/corr_otf.hip; the values of the volume lookup up to
        # fp16 rounding) unless lowmem_corr == "alt" asks for the literal fp32 alt_cuda_corr formulation
        if getattr(self, "lowmem_corr", "otf") == "otf" and str(self.device).startswith("cuda"):
            corr_op = self._otf_block()
        else:
            corr_op = AltCorrBlock(self.video.fmaps.view(1, num * rig, ch, ht, wd))
        hjj = self._host(self.jj)
        hii = self._host(self.ii)
        # The reference walks the source frames 8 at a time to bound the activations of the update operator
        # (factor_graph.py:279).  All edges of a source frame share a chunk, so the chunk size does not change
        # any value (GraphAgg averages per source frame); with 288 GB of HBM one chunk takes the whole graph
        # unless lowmem_chunk asks for less: 16x fewer launches at 128 keyframes.  Everything about a chunk
        # that does not depend on the iteration (edge selection, frame list, context features) is set up once.
        s = int(getattr(self, "lowmem_chunk", 1 << 30))
        # ... and within what the kernels can address: the convolutions use 31-bit buffer offsets, i.e. at most
        # 2^31 / (2 * 448) = 2.39 M pixel rows of the widest per-edge fp16 map per launch
        max_edges = max(1, (9 << 18) // (ht * wd))
        per_frame = np.bincount(hii, minlength=int(hii.max()) + 1) if hii.size else np.zeros(1, np.int64)
        bounds, lo, acc = [], 0, 0
        for f in range(len(per_frame)):
            if f > lo and (f - lo >= s or acc + per_frame[f] > max_edges):
                bounds.append((lo, f))
                lo, acc = f, 0
            acc += int(per_frame[f])
        bounds.append((lo, len(per_frame)))
        chunks = []
        for i, i_end in bounds:
            vh = (hii >= i) & (hii < i_end)
            if vh.sum() < 1:
                continue
            v = slice(None) if bool(vh.all()) else torch.as_tensor(vh, device=self.device)
            iis, jjs = self.ii[v], self.jj[v]
            uq_c, ix_c = torch.unique(iis, sorted=True, return_inverse=True)
            # (the per-edge context tensor of the reference call is only gathered when the operator needs it: with the
            # shared-context form the gate term is evaluated per source keyframe straight from video.inps)
            share = self.fast_update is not None and self.share_context
            chunks.append((v, iis, jjs, rig * iis, rig * jjs + (iis == jjs).long(), uq_c,
                           None if share else self.video.inps[None, iis], (uq_c, ix_c) if share else None))
        if self.fast_update is not None and self._lowmem_update is None:
            self._lowmem_update = [FusedUpdate(self.update_op)]
        while self.fast_update is not None and len(self._lowmem_update) < min(len(chunks), 4):
            self._lowmem_update.append(FusedUpdate(self.update_op))     # own context cache per chunk
        for step in range(steps):
            coords1, mask = self.video.reproject(self.ii, self.jj)
            motn = self._motion(coords1)
            for ci, (v, iis, jjs, ci1, cj1, uq, inp, ctx) in enumerate(chunks):
                whole = isinstance(v, slice)
                corr1 = corr_op(coords1 if whole else coords1[:, v], ci1, cj1)
                net_in = self.net if whole else self.net[:, v]
                mot_in = motn if whole else motn[:, v].contiguous()
                if self.fast_update is not None:
                    # chunks differ in size: separate FusedUpdate objects (own buffers, never in place); the
                    # first four chunks keep their hoisted context term across the steps
                    fu = self._lowmem_update[min(ci, len(self._lowmem_update) - 1)]
                    net, delta, weight, damping, upmask = fu(
                        net_in, inp, corr1, mot_in, iis, jjs,
                        context=(self.video.inps, ctx[0], ctx[1]) if ctx is not None else None)
                    self.video.upsample(uq, upmask, softmax_f32=True)   # inside autocast in the reference: fp32 softmax
                else:
                    with torch.autocast("cuda", enabled=True):
                        net, delta, weight, damping, upmask = \
                            self.update_op(net_in, inp, corr1, mot_in, iis, jjs)
                    self.video.upsample(uq, upmask, softmax_f32=True)
                if whole:
                    self.net = net if net.dtype == self.net.dtype else net.to(self.net.dtype)
                    self.target = coords1 + delta.float()
                    self.weight = weight.float()
                else:
                    self.net[:, v] = net
                    self.target[:, v] = coords1[:, v] + delta.float()
                    self.weight[:, v] = weight.float()
                self.damping[uq] = damping.to(self.damping.dtype)
            damping = .2 * self.damping[self._unique_ii()].contiguous() + EP
            opt_type = ("pose_depth" if step % 2 == 0 else "depth_scale") if enable_wq else "pose_depth"
            self.video.ba(self.target, self.weight, damping, self.ii, self.jj, t0, t1, iters=itrs,
                          lm=1e-5, ep=1e-2, motion_only=False, opt_type=opt_type)

    # ---- topology --------------------------------------------------------------------
    def add_neighborhood_factors(self, t0, t1, r=3):
        ii, jj = torch.meshgrid(torch.arange(t0, t1), torch.arange(t0, t1), indexing="ij")
        ii = ii.reshape(-1).to(dtype=torch.long, device=self.device)
        jj = jj.reshape(-1).to(dtype=torch.long, device=self.device)
        keep = ((ii - jj).abs() > 0) & ((ii - jj).abs() <= r)
        self.add_factors(ii[keep], jj[keep])

    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False):
        """factor_graph.py:323-383 -- distance-sorted edge proposal with NMS suppression"""
        t = self.video.counter.value
        ii, jj = np.meshgrid(np.arange(t0, t), np.arange(t1, t), indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        d = self.video.distance(ii, jj, beta=beta).cpu().numpy().astype(np.float32)
        d[ii - rad < jj] = np.inf
        d[d > 100] = np.inf
        wj = t - t1

        def suppress(i, j):
            lim = max(min(abs(i - j) - 2, nms), 0)
            for di in range(-nms, nms + 1):
                for dj in range(-nms, nms + 1):
                    if abs(di) + abs(dj) <= lim:
                        i1, j1 = i + di, j + dj
                        if t0 <= i1 < t and t1 <= j1 < t:
                            d[(i1 - t0) * wj + (j1 - t1)] = np.inf

        ii1 = np.concatenate([self._host(self.ii), self._host(self.ii_bad), self._host(self.ii_inac)])
        jj1 = np.concatenate([self._host(self.jj), self._host(self.jj_bad), self._host(self.jj_inac)])
        for i, j in zip(ii1.tolist(), jj1.tolist()):
            suppress(i, j)
        es = []
        for i in range(t0, t):
            for j in range(max(i - rad - 1, 0), i):
                es.append((i, j))
                es.append((j, i))
                d[(i - t0) * wj + (j - t1)] = np.inf
        # equal distances keep their index order: the order of the reference's device sort (cub radix sort)
        for k in np.argsort(d, kind="stable"):
            if d[k] > thresh:
                continue
            if len(es) > self.max_factors:
                break
            i, j = int(ii[k]), int(jj[k])
            es.append((i, j))
            es.append((j, i))
            suppress(i, j)
        ei, ej = torch.as_tensor(es, device=self.device).unbind(dim=-1)
        self.add_factors(ei, ej, remove)

    def add_backend_proximity_factors(self, t_start, t_end, nms, radius, thresh, max_factors, beta,
                                      t_start_loop=None, loop=False):
        """factor_graph.py:386-462"""
        if t_start_loop is None or not loop:
            t_start_loop = t_start
        assert t_start_loop >= t_start, f'short: {t_start_loop}, long: {t_start}.'
        ilen, jlen = t_end - t_start_loop, t_end - t_start
        ii, jj = np.meshgrid(np.arange(t_start_loop, t_end), np.arange(t_start, t_end), indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        d = self.video.distance(ii, jj, beta=beta).cpu().numpy().astype(np.float32)
        rawd = d.copy().reshape(ilen, jlen)
        d[ii - radius < jj] = np.inf
        d[d > thresh] = np.inf
        d = d.reshape(ilen, jlen)
        es = []
        for i in range(t_start_loop, t_end):
            for j in range(max(i - radius - 1, 0), i):
                es.append((i, j))
                es.append((j, i))
                d[i - t_start_loop, j - t_start] = np.inf
        flat = d.reshape(-1)
        order = np.argsort(flat, kind="stable")
        order = order[flat[order] <= thresh]
        loop_edges = 0
        nn = 1
        for k in order.tolist():
            di, dj = k // jlen, k % jlen
            if d[di, dj] > thresh:
                continue
            if len(es) > max_factors:
                break
            i, j = int(ii[k]), int(jj[k])
            if loop:
                sub = []
                for si in range(max(i - nn, t_start_loop), min(i + nn + 1, t_end)):
                    for sj in range(max(j - nn, t_start), min(j + nn + 1, t_end)):
                        if rawd[si - t_start_loop, sj - t_start] <= thresh and si != sj and si - sj > 20:
                            sub.append((si, sj))
                es += sub
                loop_edges += len(sub)
            else:
                es += [(i, j), (j, i)]
            d[max(0, di - nms):min(ilen, di + nms + 1), max(0, dj - nms):min(jlen, dj + nms + 1)] = np.inf
        if len(es) < 3 or (loop and loop_edges == 0):
            return 0
        ei, ej = torch.tensor(es, device=self.device).unbind(dim=-1)
        self.add_factors(ei, ej, remove=True)
        return len(self.ii)
