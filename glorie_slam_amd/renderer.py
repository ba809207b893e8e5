"""Renderer: per-ray sampling + decoder evaluation + compositing (scope rows R4-R7) -- mirror
of /root/reference/src/utils/Renderer.py (`render_batch_ray`, `eval_points`, `render_img`).

The reference evaluates 3000 rays per call to stay inside faiss limits (Renderer.py:7,267);
here the whole image goes through in `ray_batch_size` = 65536-ray chunks by default.
"""
import warnings

import torch

from . import point_ops

from .common import get_rays, raw2outputs_nerf_color


class Renderer(object):
    def __init__(self, cfg, slam, points_batch_size=2 ** 22, ray_batch_size=65536):
        self.ray_batch_size = ray_batch_size
        self.points_batch_size = points_batch_size
        r = cfg['rendering']
        self.N_surface = r['N_surface']
        self.near_end_surface = r['near_end_surface']
        self.far_end_surface = r['far_end_surface']
        self.sample_near_pcl = r['sample_near_pcl']
        self.sigmoid_coefficient = r['sigmoid_coef']
        self.near_end = r['near_end']
        self.use_dynamic_radius = cfg['pointcloud']['use_dynamic_radius']
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = slam.H, slam.W, slam.fx, slam.fy, slam.cx, slam.cy

    def eval_points(self, p, decoders, npc, stage='color', device=None, npc_geo_feats=None,
                    npc_col_feats=None, is_tracker=False, cloud_pos=None, pts_views_d=None,
                    ray_pts_num=None, dynamic_r_query=None):
        rets, ray_masks, point_masks, counters = [], [], [], []
        bs = self.points_batch_size - self.points_batch_size % max(ray_pts_num or 1, 1)
        for s in range(0, p.shape[0], bs):
            sl = slice(s, s + bs)
            ret, vm, pm, cnt = decoders(p[sl].unsqueeze(0), npc, stage, npc_geo_feats, npc_col_feats,
                                        ray_pts_num, is_tracker, cloud_pos,
                                        pts_views_d[sl] if pts_views_d is not None else None,
                                        dynamic_r_query[sl] if dynamic_r_query is not None else None)
            ret = ret.squeeze(0)
            if ret.dim() == 1 and ret.shape[0] == 4:
                ret = ret.unsqueeze(0)
            rets.append(ret); ray_masks.append(vm); point_masks.append(pm); counters.append(cnt)
        return torch.cat(rets, 0), torch.cat(ray_masks, 0), torch.cat(point_masks, 0), torch.cat(counters)

    def sample_z(self, npc, rays_o, rays_d, gt_depth, device):
        """z_vals [R,S] and the near-cloud mask (Renderer.py:106-174)"""
        S = self.N_surface
        R = rays_o.shape[0]
        if gt_depth is not None:
            far = torch.minimum(5 * gt_depth.mean(), torch.max(gt_depth * 1.2)).repeat(R, 1).float()
            if torch.numel(gt_depth) != 0:
                gt_depth = gt_depth.reshape(-1, 1)
            else:
                warnings.warn('tensor gt_depth is empty')
                gt_depth = torch.zeros(R, 1, device=device)
        else:
            far = 10 * torch.ones((R, 1), device=device).float()
            gt_depth = torch.zeros(R, 1, device=device)
        nz = (gt_depth > 0).squeeze(-1)
        near_mask = torch.ones(R, device=device, dtype=torch.bool)
        t = torch.linspace(0.0, 1.0, steps=S, device=device)
        z = self.near_end_surface * gt_depth * (1. - t) + self.far_end_surface * gt_depth * t
        z = torch.where(nz[:, None], z, torch.zeros_like(z))
        if int(nz.sum()) < R:
            if self.sample_near_pcl:
                z0, not_near = npc.sample_near_pcl(rays_o[~nz].detach().clone(), rays_d[~nz].detach().clone(),
                                                   self.near_end, torch.max(far), S)
                if torch.sum(not_near.ravel()):
                    near_mask[torch.nonzero(~nz, as_tuple=True)[0][not_near]] = False
                z[~nz] = z0
            else:
                z[~nz] = torch.linspace(self.near_end, torch.max(far), steps=S, device=device).repeat((~nz).sum(), 1)
        return z, near_mask, nz

    def _render_fast(self, npc, decoders, rays_d, rays_o, stage, gt_depth, npc_geo_feats, npc_col_feats,
                     cloud_pos, dynamic_r_query, image_w=None, camera=None, precise=False):
        """Inference path of render_batch_ray for batches in which every ray has a depth prior: seven HIP
        launches (samples, KNN, IDW gather, three decoders, per-ray counts, compositing) and no torch glue.
        Returns None when a ray has no depth (sample_near_pcl is needed: general path)."""
        S = self.N_surface
        R = gt_depth.shape[0]
        g = decoders.geo_decoder
        rad = dynamic_r_query if self.use_dynamic_radius else None
        if camera is not None:
            # render_img: the rays of this strip are formed inside the sampling kernel (get_rays fused, row R7)
            cam, first_pixel = camera
            z_vals, pts, views, rq, n_zero = point_ops.ray_samples_camera(cam, image_w, first_pixel, gt_depth, rad, S,
                                                                          self.near_end_surface, self.far_end_surface)
        else:
            z_vals, pts, views, rq, n_zero = point_ops.ray_samples(rays_o, rays_d, gt_depth, rad, S,
                                                                   self.near_end_surface, self.far_end_surface)
        # the zero-depth count travels to pinned host memory behind the sampling kernel; it is looked at after
        # the rest of the batch has been enqueued, so the device never waits for the host
        flag = self._pinned_flag()
        flag.copy_(n_zero, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        # neighbours + IDW weights + mask from ONE launch (the geometry kernel interpolates its feature itself: no [Q,32] round
        # trip).  The search is bounded by the query radius (per sample, or the cloud's fixed one): I / D are exact inside
        # the ball - the only slots with a non-zero weight - and unspecified beyond it (NeuralPointCloud.find_neighbors_faiss)
        expo = getattr(g, "weighting", "distance") != "distance"
        D, I, nn_num, w, has = npc.find_neighbors_faiss(pts, step='query', dynamic_radius=rq,
                                                        image_layout=(S, image_w) if image_w else None,
                                                        weights=(g.min_nn_num, expo, True))
        cp = cloud_pos if cloud_pos is not None else npc.cloud_pos()
        # (precise: exact-fp32 decoder kernels - the caller's answer to a tripped range guard of the fp16-split ones)
        raw = point_ops.render_mlp(decoders._packed(), pts, views, cp, npc_col_feats, None, I, w, has,
                                   stage=stage, geo_feats=npc_geo_feats, precise=precise,
                                   range_flag=None if precise else decoders.range_guard(pts.device).flag)
        counts, valid = point_ops.ray_counts(has, S, 3)
        depth, var, rgb, _ = point_ops.composite(raw.view(R, S, 4), z_vals, self.sigmoid_coefficient,
                                                 return_weights=False)
        ev.synchronize()
        if int(flag[0]) != 0:
            return None
        return depth, var, rgb, valid, counts

    def _pinned_flag(self):
        # one word per batch stream (batch_stream): two batches can be in flight
        if getattr(self, "_flags", None) is None:
            self._flags = torch.zeros(2, dtype=torch.int32).pin_memory()
        k = getattr(self, "_flag_slot", 0)
        return self._flags[k:k + 1]

    def batch_stream(self, k, device):
        """stream of a frame's k-th batch (not in the reference): the batches of a frame are independent, and on alternating
        streams the latency-bound neighbour search of batch k + 1 runs beside the matrix-bound decoders of batch k (-3...4 % on
        a 640x480 frame, same values).  Usage (render_img, bench.render_pass): `with torch.cuda.stream(ren.batch_stream(k, dev)):
        render_batch_ray(...)` for every batch, then `ren.join_batches(dev)` before the results are read."""
        dev = torch.device(device)
        if dev.index is None:                  # 'cuda' names the current device
            dev = torch.device("cuda", torch.cuda.current_device())
        st = getattr(self, "_streams", None)
        if st is None or st[0].device != dev:
            st = self._streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        if k < 2:
            st[k].wait_stream(torch.cuda.current_stream(dev))      # the frame's inputs were produced on the caller's stream
        self._flag_slot = k & 1
        return st[k & 1]

    def prepare_frame(self, npc, decoders, device):
        """Everything the batches of a frame SHARE and build lazily is materialised here, on the caller's stream, before the
        first `batch_stream` fork: the decoders' packed parameters (`pack_decoders`: dozens of asynchronous cat / copy kernels,
        again after every FeatureAdam.step), their range-guard word, the sample fractions and the cloud's position table.
        Otherwise batch 0 builds them on stream 0 while batch 1 - a cache hit on stream 1, ordered only against the caller's
        stream - may launch `render_mlp` on a half-written pack or an unzeroed flag."""
        dev = torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        decoders._packed()
        decoders.range_guard(dev)
        point_ops.t_lin(dev, self.N_surface)
        if npc is not None:
            npc.cloud_pos()

    def join_batches(self, device):
        st = getattr(self, "_streams", None)
        if st is not None:
            for s_ in st:
                torch.cuda.current_stream(torch.device(device)).wait_stream(s_)
        self._flag_slot = 0

    def _fast_ok(self, decoders, probe, R, gt_depth, stage, npc_geo_feats, npc_col_feats, is_tracker, dynamic_r_query):
        """the batch can take the all-HIP inference path (every ray must also have a depth prior: checked on the device)"""
        return (gt_depth is not None and R > 0 and torch.numel(gt_depth) == R and stage in ('geometry', 'color')
                and getattr(self, "use_fast_path", True)
                and (dynamic_r_query is not None or not self.use_dynamic_radius)
                and decoders._fused_ok(probe, npc_geo_feats, npc_col_feats, is_tracker, stage)
                and decoders.geo_decoder.use_dynamic_radius == self.use_dynamic_radius)

    def render_batch_ray(self, npc, decoders, rays_d, rays_o, device, stage, gt_depth=None,
                         npc_geo_feats=None, npc_col_feats=None, is_tracker=False, cloud_pos=None,
                         dynamic_r_query=None, image_w=None, defer_guard=False):
        """Renderer.py:80-219 -> depth, uncertainty, color, valid_ray_mask, valid_ray_counts
        image_w (not in the reference): the rays are consecutive row-major pixels of an image of that width, starting
        at a row start - lets the neighbour search walk the image in patches; the result does not depend on it.
        defer_guard (not in the reference): do not read the decoders' range guard behind this batch (a host synchronisation:
        the device then idles while the next batch is enqueued, ~0.1 ms per batch) - the caller reads
        `decoders.range_guard(device).tripped()` once behind its last batch and, if it is set, renders again without this flag
        (what render_img does for its strips)."""
        S = self.N_surface
        R = rays_o.shape[0]
        if self._fast_ok(decoders, rays_o, R, gt_depth, stage, npc_geo_feats, npc_col_feats, is_tracker, dynamic_r_query):
            out = self._render_fast(npc, decoders, rays_d, rays_o, stage, gt_depth, npc_geo_feats,
                                    npc_col_feats, cloud_pos, dynamic_r_query, image_w=image_w)
            if out is not None and not defer_guard and decoders.range_guard(rays_o.device).tripped():
                # an activation, feature or weight outside the fp16 range: the split kernels' result is void
                out = self._render_fast(npc, decoders, rays_d, rays_o, stage, gt_depth, npc_geo_feats,
                                        npc_col_feats, cloud_pos, dynamic_r_query, image_w=image_w, precise=True)
            if out is not None:
                return out
        if (torch.is_grad_enabled() and gt_depth is not None and R > 0 and torch.numel(gt_depth) == R and not is_tracker
                and stage in ('geometry', 'color') and rays_o.is_cuda and getattr(self, "use_train_path", True)
                and (dynamic_r_query is not None or not self.use_dynamic_radius)
                and decoders.geo_decoder.use_dynamic_radius == self.use_dynamic_radius):
            # mapping iteration (mapper.py:390-515): the same kernels chain under autograd - gradients with respect to
            # the feature tables and the decoder parameters come from csrc/train.hip, not from ~150 torch ops
            from . import render_train
            wants = (npc_geo_feats.requires_grad or npc_col_feats.requires_grad
                     or any(p.requires_grad for p in decoders.parameters()))
            if wants and render_train.supported(decoders):
                out = render_train.render_rays(self, npc, decoders, rays_d, rays_o, stage, gt_depth, npc_geo_feats,
                                               npc_col_feats, cloud_pos, dynamic_r_query)
                if out is not None:
                    return out
        z_vals, near_mask, nz = self.sample_z(npc, rays_o, rays_d, gt_depth, device)
        pts = (rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]).reshape(-1, 3)
        rays_d_pts = rays_d.repeat_interleave(S, dim=0).reshape(-1, 3)
        if self.use_dynamic_radius:
            dynamic_r_query = dynamic_r_query.reshape(-1, 1).repeat_interleave(S, dim=0)
        raw, valid_ray_mask, point_mask, counts = self.eval_points(
            pts, decoders, npc, stage, device, npc_geo_feats, npc_col_feats, is_tracker, cloud_pos,
            rays_d_pts, ray_pts_num=S, dynamic_r_query=dynamic_r_query)
        with torch.no_grad():
            raw[torch.nonzero(~point_mask).flatten(), -1] = -100.0
        raw = raw.reshape(R, S, -1)
        depth, uncertainty, color, _ = raw2outputs_nerf_color(raw, z_vals, rays_d, device=device,
                                                              coef=self.sigmoid_coefficient)
        valid_ray_mask = valid_ray_mask & near_mask
        if not self.sample_near_pcl:
            depth[~nz] = 0
        return depth, uncertainty, color, valid_ray_mask, counts

    @torch.no_grad()
    def render_img(self, npc, decoders, c2w, device, stage, gt_depth=None, npc_geo_feats=None,
                   npc_col_feats=None, dynamic_r_query=None, cloud_pos=None, _precise=False):
        """Renderer.py:221-306"""
        H, W = self.H, self.W
        n_rays = H * W
        if self.use_dynamic_radius:
            dynamic_r_query = dynamic_r_query.reshape(-1, 1)
        gt = gt_depth.reshape(-1) if gt_depth is not None else None
        outs = [[], [], [], [], []]
        bs = self.ray_batch_size
        if bs >= 16 * W:
            bs -= bs % (16 * W)          # whole 16-row strips: every batch starts at a row start (image_w hint below)
        image_w = W if bs % W == 0 else None
        # strips in which every pixel has a depth prior never materialise their rays: get_rays (common.py:302-322) is
        # evaluated inside the sampling kernel (row R7 fused into R4); the ray tensors are only built if a strip needs
        # the general path
        cam = point_ops.camera_block(c2w, self.fx, self.fy, self.cx, self.cy, device) \
            if (image_w and gt is not None and gt.is_cuda and getattr(self, "fuse_get_rays", True)) else None
        rays = None
        bad_any = False
        two = cam is not None and n_rays > bs and not torch.is_grad_enabled()
        if two:
            self.prepare_frame(npc, decoders, device)
        for k, i in enumerate(range(0, n_rays, bs)):
            g_i = gt[i:i + bs] if gt is not None else None
            r_i = dynamic_r_query[i:i + bs] if self.use_dynamic_radius else None
            ret = None
            if cam is not None and self._fast_ok(decoders, g_i, g_i.shape[0], g_i, stage, npc_geo_feats, npc_col_feats,
                                                 False, r_i):
                with torch.cuda.stream(self.batch_stream(k, device) if two else torch.cuda.current_stream(torch.device(device))):
                    ret = self._render_fast(npc, decoders, None, None, stage, g_i, npc_geo_feats, npc_col_feats, cloud_pos,
                                            r_i, image_w=image_w, camera=(cam, i), precise=_precise)
            if ret is None and two:
                self.join_batches(device)              # the general path runs on the caller's stream
                two = False
            if ret is None:
                if cam is not None and not _precise:
                    # the general path looks at (and clears) the range guard itself: keep what the strips so far raised
                    bad_any = bad_any or decoders.range_guard(device).tripped()
                if rays is None:
                    rays_o, rays_d = get_rays(H, W, self.fx, self.fy, self.cx, self.cy, c2w, device)
                    rays = (rays_o.reshape(-1, 3), rays_d.reshape(-1, 3))
                ret = self.render_batch_ray(
                    npc, decoders, rays[1][i:i + bs], rays[0][i:i + bs], device, stage, gt_depth=g_i,
                    npc_geo_feats=npc_geo_feats, npc_col_feats=npc_col_feats, cloud_pos=cloud_pos,
                    dynamic_r_query=r_i, image_w=image_w)
            for o, v in zip(outs, ret):
                o.append(v)
        if two:
            self.join_batches(device)
        if cam is not None and not _precise and (decoders.range_guard(device).tripped() or bad_any):
            # the range guard of the fp16-split decoders is looked at once per frame (no stall between the strips): a trip
            # voids the frame, which is rendered again on the exact-fp32 kernels
            return self.render_img(npc, decoders, c2w, device, stage, gt_depth=gt_depth, npc_geo_feats=npc_geo_feats,
                                   npc_col_feats=npc_col_feats,
                                   dynamic_r_query=dynamic_r_query.reshape(H, W) if self.use_dynamic_radius else None,
                                   cloud_pos=cloud_pos, _precise=True)
        depth = torch.cat(outs[0]).double().reshape(H, W)
        unc = torch.cat(outs[1]).double().reshape(H, W)
        color = torch.cat(outs[2]).reshape(H, W, 3)
        vmask = torch.cat(outs[3]).double().reshape(H, W)
        vcount = torch.cat(outs[4]).double().reshape(H, W)
        return depth, unc, color, vmask, vcount
