"""NeuralPointCloud: point store + neighbour search (scope row R1) -- mirror of the query
side of /root/reference/src/neural_point.py.  The faiss IndexIVFFlat is replaced by the
exact cell-list search of libglorie_hip (`point_ops.KnnIndex`); `find_neighbors_faiss` and
`sample_near_pcl` keep their signatures and return conventions.

Point-cloud maintenance: `add_neural_points` (radius-test insertion
along the rays of a keyframe, neural_point.py:165-262), `add_points(video_idxs)` (full-resolution
unprojection of keyframes, :145-162), `update_points_pos` (deformation after a pose/depth update,
:378-438) and `retrain_updated_points`.  Every index (re)build is one counting sort on the device
(`glorie_knn_build`), where the reference re-trains an IVF k-means.  The module-level functions of the
reference follow at the end: `proj_depth_map` (one z-buffer launch), `get_proxy_render_depth`,
`update_points_pos(npc, video)` (:446-575).
"""
import os

import numpy as np
import torch

from . import _lib as L
from . import point_ops


def get_scale(depth_prev, depth_curr):
    """neural_point.py:11-16: least-squares s with depth_prev * s = depth_curr"""
    return torch.sum(depth_prev * depth_curr) / torch.sum(depth_prev * depth_prev)


def se3_inv(poses):
    """inverse of [tx ty tz qx qy qz qw] rigid transforms (lietorch SE3(poses).inv().data)"""
    t, q = poses[..., :3], poses[..., 3:]
    qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
    # rotate -t by qi:  v + 2 w (u x v) + 2 u x (u x v),  u = qi.xyz, w = qi.w
    u, w = qi[..., :3], qi[..., 3:]
    v = -t
    uv = torch.cross(u, v, dim=-1)
    return torch.cat([v + 2 * (w * uv + torch.cross(u, uv, dim=-1)), qi], -1).contiguous()


class NeuralPointCloud(object):
    def __init__(self, cfg, video=None):
        self.cfg = cfg
        self.c_dim = cfg['model']['c_dim']
        self.device = cfg['device']
        pc = cfg['pointcloud']
        self.use_dynamic_radius = pc['use_dynamic_radius']
        self.nn_num = pc['nn_num']
        self.nlist = pc.get('nlist', 400)      # faiss parameters: accepted, unused (exact search)
        self.radius_add = pc['radius_add']
        self.radius_min = pc['radius_min']
        self.radius_query = pc['radius_query']
        self.N_add = pc.get('N_add', 3)
        self.near_end_surface = pc.get('near_end_surface', 0.95)
        self.far_end_surface = pc.get('far_end_surface', 1.05)
        self.fix_interval_when_add_along_ray = pc.get('fix_interval_when_add_along_ray', False)
        self._cloud_pos = None       # (input_pos) * N_add
        self._input_pos = None       # locations of the depth inputs the cloud was seeded from
        self._input_rgb = None
        self._input_video_idx = None
        self._input_j = None         # pixel of the input, depth[j, i]
        self._input_i = None
        self._input_depth = None
        self._full_pcl = None        # [buffer,H,W,3] unprojected keyframes (allocated by add_points(video_idxs))
        self._full_mask = None
        self._pts_num = 0
        self.geo_feats = None
        self.col_feats = None
        self.video = video
        self.index = point_ops.KnnIndex(self.device, cell_size=pc.get('knn_cell_size', 0.06),
                                        max_cells=pc.get('knn_max_cells', 1 << 21))

    # ---- accessors (same names as the reference) --------------------------------------
    def get_device(self):
        return self.device

    def cloud_pos(self, index=None):
        return self._cloud_pos if index is None else self._cloud_pos[index]

    def pts_num(self):
        return self._pts_num

    def input_pos(self):
        return self._input_pos

    def input_rgb(self):
        return self._input_rgb

    def input_i(self):
        return self._input_i

    def input_j(self):
        return self._input_j

    def input_video_idx(self):
        return self._input_video_idx

    def full_pcl(self):
        return self._full_pcl

    def full_mask(self):
        return self._full_mask

    def get_N_add(self):
        return self.N_add

    def get_near_end_surface(self):
        return self.near_end_surface

    def get_far_end_surface(self):
        return self.far_end_surface

    def is_fix_interval_when_add_along_ray(self):
        return self.fix_interval_when_add_along_ray

    def get_radius_query(self):
        return self.radius_query

    def get_geo_feats(self):
        return self.geo_feats

    def get_col_feats(self):
        return self.col_feats

    def index_train(self, xb):
        self.index.train(xb)
        return True

    def index_reset(self):
        self.index.reset()

    def index_add(self, xb):
        self.index.add(xb)

    def index_ntotal(self):
        return self.index.ntotal

    def update_geo_feats(self, feats, indices=None):
        if indices is not None:
            self.geo_feats[indices] = feats.detach().clone()
        else:
            assert feats.shape[0] == self.geo_feats.shape[0], 'feature shape[0] mismatch'
            self.geo_feats = feats.detach().clone()

    def update_col_feats(self, feats, indices=None):
        if indices is not None:
            self.col_feats[indices] = feats.detach().clone()
        else:
            assert feats.shape[0] == self.col_feats.shape[0], 'feature shape[0] mismatch'
            self.col_feats = feats.detach().clone()

    # ---- store ------------------------------------------------------------------------
    def add_points(self, video_idxs, geo_feats=None, col_feats=None):
        """Two call forms.
        add_points(video_idxs: int | int64 tensor) -- the reference's method (neural_point.py:145-162):
            unproject the full-resolution depth of those keyframes of `self.video` into
            `_full_pcl` / `_full_mask`; returns the number of valid pixels.
        add_points(pts [n,3] float, geo_feats=None, col_feats=None) -- plain append (+ features,
            default N(0,0.1) as neural_point.py:243-246) and rebuild of the search structure."""
        pts = video_idxs                                   # the parameter keeps the reference's name (keyword callers)
        if isinstance(pts, int) or (torch.is_tensor(pts) and not pts.is_floating_point()):
            return self._add_video_points(pts)
        pts = pts.detach().to(self.device, torch.float32).reshape(-1, 3)
        n = pts.shape[0]
        mk = lambda f: f.detach().to(self.device, torch.float32) if f is not None else \
            torch.zeros([n, self.c_dim], device=self.device).normal_(mean=0, std=0.1)
        g, c = mk(geo_feats), mk(col_feats)
        if self._cloud_pos is None:
            self._cloud_pos, self.geo_feats, self.col_feats = pts.clone(), g, c
        else:
            self._cloud_pos = torch.cat([self._cloud_pos, pts])
            self.geo_feats = torch.cat([self.geo_feats, g], 0)
            self.col_feats = torch.cat([self.col_feats, c], 0)
        self._pts_num = self._cloud_pos.shape[0]
        self.index.set_points(self._cloud_pos)
        return n

    def _add_video_points(self, video_idxs):
        from . import droid_backends
        v = self.video
        if isinstance(video_idxs, int):
            video_idxs = torch.tensor([video_idxs], dtype=torch.long, device=self.device)
        if self._full_pcl is None:
            B, H, W = v.fresh_disps_up().shape
            self._full_pcl = torch.zeros(B, H, W, 3, device=self.device, dtype=torch.float)
            self._full_mask = torch.zeros(B, H, W, device=self.device, dtype=torch.bool)
        intrinsic = (v.intrinsics[0].detach() * float(v.down_scale)).contiguous()
        masks = torch.index_select(v.valid_depth_mask.detach(), 0, video_idxs)
        disps = torch.index_select(v.disps_up.detach(), 0, video_idxs).contiguous()
        poses = torch.index_select(v.poses.detach(), 0, video_idxs).contiguous()
        self._full_pcl[video_idxs] = droid_backends.iproj(se3_inv(poses), disps, intrinsic)   # SE3(poses).inv().data
        self._full_mask[video_idxs] = masks
        return torch.sum(masks)

    def _z_along_ray(self, depth):
        """the N_add sample depths around a surface depth (neural_point.py:215-229)"""
        d = depth.unsqueeze(-1).repeat(1, self.N_add)
        if self.fix_interval_when_add_along_ray:
            return d + torch.linspace(-0.04, 0.04, steps=self.N_add, device=self.device).unsqueeze(0)
        t = torch.linspace(0.0, 1.0, steps=self.N_add, device=self.device)
        return self.near_end_surface * d * (1. - t) + self.far_end_surface * d * t

    @torch.no_grad()
    def add_neural_points(self, batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color,
                          video_idx, i, j, train=False, is_pts_grad=False, dynamic_radius=None):
        """neural_point.py:165-262: seed N_add neural points along every ray whose surface point has no
        neighbour within radius_add (radius_min / the per-ray dynamic radius) in the current cloud.
        Returns the number of accepted rays.  `train` is accepted for signature compatibility: the cell
        list is rebuilt exactly on every insertion."""
        if not batch_rays_o.shape[0]:
            return 0
        mask = batch_gt_depth > 0
        mask = mask * (batch_gt_depth < batch_gt_depth.quantile(0.8) * 2.0)
        batch_gt_color = batch_gt_color * 255
        batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color = \
            batch_rays_o[mask], batch_rays_d[mask], batch_gt_depth[mask], batch_gt_color[mask]
        i, j = i[mask], j[mask]
        if dynamic_radius is not None:
            dynamic_radius = dynamic_radius[mask]
        pts_gt = (batch_rays_o[..., None, :] + batch_rays_d[..., None, :] * batch_gt_depth[..., None, None]).reshape(-1, 3)
        mask = torch.ones(pts_gt.shape[0], device=self.device).bool()
        if self.index.ntotal > 0 and pts_gt.shape[0]:      # (reference: `index.is_trained`, i.e. a non-empty cloud)
            # only the COUNT inside the radius is consumed: the search is bounded by the ball (weights = (.., ball_only)), which
            # leaves that count exact.  The unbounded exact 8-NN search expands shells until eight neighbours are found - for
            # the points of a newly seen surface, far from the cloud, that walked large parts of the grid: 2.5-5.4 ms per
            # keyframe insertion on the long synthetic sequence (tools/exp_naive.py, round 6) against ~0.1 ms bounded
            neighbor_num_gt = self.find_neighbors_faiss(pts_gt, step='add', is_pts_grad=is_pts_grad,
                                                        dynamic_radius=dynamic_radius, weights=(1, False, True))[2]
            mask = (neighbor_num_gt == 0)
        new = dict(_input_pos=pts_gt[mask], _input_rgb=batch_gt_color[mask], _input_depth=batch_gt_depth[mask],
                   _input_video_idx=video_idx * torch.ones_like(i[mask], dtype=torch.long, device=self.device),
                   _input_i=i[mask], _input_j=j[mask])
        for name, val in new.items():
            val = val.detach().clone()
            cur = getattr(self, name)
            setattr(self, name, val if cur is None else torch.cat([cur, val]))
        pts = batch_rays_o[..., None, :] + batch_rays_d[..., None, :] * self._z_along_ray(batch_gt_depth)[..., :, None]
        pts = pts[mask].reshape(-1, 3)                 # auxiliary points share the mask of the surface point
        if pts.shape[0]:
            NeuralPointCloud.add_points(self, pts)     # append + N(0,0.1) features + rebuild of the cell list
        return torch.sum(mask)

    @torch.no_grad()
    def update_points_pos(self, v_idx, depth, c2w, cfg):
        """neural_point.py:378-438: move the points that were unprojected from keyframe v_idx after its
        depth map / pose changed (call retrain_updated_points afterwards, like the reference)"""
        from .common import get_rays_from_uv, update_cam
        depth = depth.to(self.device)
        frame_mask = (self._input_video_idx == v_idx)
        if frame_mask.sum() == 0:
            return
        pj, pi = self._input_j[frame_mask], self._input_i[frame_mask]
        depth_prev = self._input_depth[frame_mask]
        pd = depth[pj, pi]
        bad = (pd == 0.0)
        if bad.sum() > 0:
            scale = torch.sum(depth_prev[~bad] * pd[~bad]) / torch.sum(depth_prev[~bad] * depth_prev[~bad])   # get_scale
            pd[bad] = scale * depth_prev[bad]
        _, _, fx, fy, cx, cy = update_cam(cfg)
        rays_o, rays_d = get_rays_from_uv(pi, pj, c2w, fx, fy, cx, cy, self.device)
        self._input_pos[frame_mask] = (rays_o[..., None, :] + rays_d[..., None, :] * pd[..., None, None]).reshape(-1, 3)
        self._input_depth[frame_mask] = pd.clone()
        pts = (rays_o[..., None, :] + rays_d[..., None, :] * self._z_along_ray(pd)[..., :, None]).reshape(-1, 3)
        self._cloud_pos[frame_mask.unsqueeze(-1).repeat(1, self.N_add).reshape(-1)] = pts

    def retrain_updated_points(self):
        """after positions changed (deformation): rebuild (neural_point.py:441-444)"""
        self.index.set_points(self._cloud_pos)

    # ---- search -----------------------------------------------------------------------
    def find_neighbors_faiss(self, pos, step='add', retrain=False, is_pts_grad=False, dynamic_radius=None,
                             image_layout=None, weights=None):
        """neural_point.py:264-313 -> (D [Q,nn] squared distances, I [Q,nn] int64, neighbor_num [Q] int32).
        image_layout: see KnnIndex.search (an ordering hint for image-shaped query batches, same result);
        weights = (min_nn, expo[, ball_only]): the IDW weights and neighbour mask of the decoders from the same launch (two more
        returns).  With ball_only (what both render paths pass) the search stops at the query radius: D / I are then the
        EXACT nearest neighbours only for the slots inside the ball (d <= r^2, the ones with a non-zero weight, counted by
        neighbor_num); slots beyond it may hold farther points, or I = -1 / D = FLT_MAX - a relaxed contract compared with
        the reference's faiss result, whose consumers (decoder.py:130-173) never read those slots either."""
        assert step in ['add', 'query']
        if retrain:
            self.index.set_points(self._cloud_pos)
        if step == 'query':
            radius = self.radius_query
        else:
            radius = self.radius_add if not is_pts_grad else self.radius_min
        if dynamic_radius is not None:
            assert pos.shape[0] == dynamic_radius.shape[0], 'shape mis-match for input points and dynamic radius'
        return self.index.search(pos, self.nn_num, radius=radius, radius_per_query=dynamic_radius,
                                 image_layout=image_layout, weights=weights)

    def sample_near_pcl(self, rays_o, rays_d, near, far, num):
        """neural_point.py:315-375: z-samples for rays without depth, bracketed by the first two
        of 25 probe samples that have a neighbour.  Returns (z_vals [n,num], invalid_mask [n])."""
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        n_rays = rays_d.shape[0]
        intervals = 25
        if torch.is_tensor(far):
            far = far.item()
        z_probe = torch.linspace(near, far, steps=intervals, device=self.device)
        pts = (rays_o[..., None, :] + rays_d[..., None, :] * z_probe[..., :, None]).reshape(-1, 3)
        nn_num = self.find_neighbors_faiss(pts, step='query', weights=(1, False, True))[2]     # (count only: bounded by the ball)
        occ = nn_num.reshape(n_rays, intervals) > 0
        invalid = occ.sum(-1) < 2
        # first and second occupied probe of every ray, then num samples between them - the reference walks the
        # rays in a numpy loop on the host (neural_point.py:355-372); same float64 linspace arithmetic here,
        # evaluated for all rays at once on the device
        z_section = torch.linspace(near, far, steps=intervals, device=self.device, dtype=torch.float64)
        first = torch.argmax(occ.to(torch.uint8), dim=1)
        occ2 = occ.clone()
        occ2[torch.arange(n_rays, device=self.device), first] = False
        second = torch.argmax(occ2.to(torch.uint8), dim=1)
        lo = torch.where(invalid, torch.full_like(z_section[first], near), z_section[first])
        hi = torch.where(invalid, torch.full_like(z_section[second], far), z_section[second])
        steps = torch.arange(num, device=self.device, dtype=torch.float64)
        z_total = lo[:, None] + ((hi - lo) / max(num - 1, 1))[:, None] * steps[None]
        if num > 1:
            z_total[:, -1] = hi                      # numpy.linspace pins the end point
        return z_total.float(), invalid



# ---- module-level helpers of the reference's neural_point.py (proxy depth, deformation driver) ----------
def proj_depth_map(c2w, npc, device, cfg, neural_pcl=False):
    """neural_point.py:446-506: depth map of the point set seen from `c2w` (closest point per pixel, 0 where
    no point lands).  neural_pcl=False projects the unprojected keyframe point maps (`npc.full_pcl()`) minus the
    frame `mapping_window_size` behind the newest one, True the neural points themselves.  One z-buffer
    launch (`glorie_proj_depth`, atomicMin) instead of a global sort + unique of all projected points."""
    from .common import update_cam
    H, W, fx, fy, cx, cy = update_cam(cfg)
    dev = npc.get_device()
    if neural_pcl:
        points, mask = npc.cloud_pos().contiguous(), None
    else:
        full_mask = npc.full_mask().clone()
        full_mask[npc.video.counter.value - cfg["mapping"]["mapping_window_size"]] = False
        points, mask = npc.full_pcl().reshape(-1, 3), full_mask.reshape(-1).view(torch.uint8)
    w2c = torch.linalg.inv(c2w.to(dev, torch.float32)).contiguous()
    depth = torch.full((H, W), float("inf"), device=dev, dtype=torch.float32)
    n = points.shape[0]
    if n:
        L.check(L.load().glorie_proj_depth(L.ptr(points), L.ptr(mask), n, L.ptr(w2c), float(fx), float(fy),
                                           float(cx), float(cy), H, W, L.ptr(depth), L.stream_ptr()),
                "glorie_proj_depth")
    return torch.where(torch.isinf(depth), torch.zeros_like(depth), depth).to(device)


def get_proxy_render_depth(npc, cfg, c2w, droid_depth, mono_depth, device, idx=None, use_mono_to_complete=True):
    """neural_point.py:539-575: the proxy depth of the paper - the tracker's depth where it is valid, else the
    depth of the point cloud projected into the view, else (optionally) the aligned mono prior"""
    proxy = droid_depth.clone()
    proj = proj_depth_map(c2w, npc, device, cfg)
    take = (~(droid_depth > 0.0)) & (proj > 0.0)
    proxy[take] = proj[take]
    if cfg["mapping"].get("save_depth", False) and idx is not None:
        out = f"{cfg['data']['output']}/{cfg['setting']}/{cfg['scene']}/semi_dense_depth"
        p_droid, p_proj = f"{out}/droid/{idx:05d}.npy", f"{out}/project/{idx:05d}.npy"
        if not os.path.isfile(p_droid):
            os.makedirs(os.path.dirname(p_droid), exist_ok=True)
            os.makedirs(os.path.dirname(p_proj), exist_ok=True)
            np.save(p_droid, droid_depth.detach().cpu().float().numpy())
            np.save(p_proj, proxy.detach().cpu().float().numpy())
    if use_mono_to_complete:
        hole = proxy == 0
        proxy[hole] = mono_depth[hole]
    return proxy


def update_points_pos(npc, video, mono_depth_loader=None):
    """neural_point.py:509-537: deform the cloud after the tracker moved keyframes - for every keyframe whose
    `npc_dirty` flag is set, re-place the points that were created from it with its current depth map and pose,
    refresh its unprojected point map and rebuild the search structure.  `mono_depth_loader(dataset_idx)` is
    only needed for cfg['mapping']['render_depth'] == 'mono' (the reference reads the prior from disk there)."""
    with video.get_lock():
        video_idx, = torch.where(video.npc_dirty.clone())
    if len(video_idx) == 0 or npc.pts_num() == 0:
        return
    video.npc_dirty[video_idx] = False
    device = npc.get_device()
    for v_idx in video_idx:
        est_depth, est_mask, c2w = video.get_depth_and_pose(v_idx, device)
        est_depth[~est_mask] = 0
        c2w[:3, 1:3] *= -1
        mode = video.cfg["mapping"]["render_depth"]
        if mode == "proxy":
            render_depth = est_depth
        elif mode == "mono":
            if mono_depth_loader is None:
                raise RuntimeError("render_depth == 'mono' needs a mono_depth_loader")
            mono = mono_depth_loader(int(video.timestamp[v_idx])).to(device)
            scale, shift = video.get_depth_scale_and_shift(v_idx, mono, est_depth, est_depth > 0)
            render_depth = mono * scale + shift
        else:
            raise NotImplementedError(mode)
        npc.update_points_pos(v_idx, render_depth.clone(), c2w.clone(), video.cfg)
    npc.add_points(video_idx)
    npc.retrain_updated_points()
