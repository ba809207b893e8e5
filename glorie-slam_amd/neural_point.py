"""NeuralPointCloud: point store + neighbour search (scope row R1) -- mirror of the query
side of /root/reference/src/neural_point.py.  The faiss IndexIVFFlat is replaced by the
exact cell-list search of libglorie_hip (`point_ops.KnnIndex`); `find_neighbors_faiss` and
`sample_near_pcl` keep their signatures and return conventions.

Point insertion driven by depth maps, deformation and proxy-depth rendering
(neural_point.py:145-262, 377-575) are the "next" row N2 of SURVEY.md section 8(f) and are not
part of this module yet; `add_points` below is the plain store/append they build on.
"""
import numpy as np
import torch

from . import point_ops


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
        self._cloud_pos = None
        self._pts_num = 0
        self.geo_feats = None
        self.col_feats = None
        self.video = video
        self.index = point_ops.KnnIndex(self.device, cell_size=pc.get('knn_cell_size', 0.08),
                                        max_cells=pc.get('knn_max_cells', 1 << 21))

    # ---- accessors (same names as the reference) --------------------------------------
    def get_device(self):
        return self.device

    def cloud_pos(self, index=None):
        return self._cloud_pos if index is None else self._cloud_pos[index]

    def pts_num(self):
        return self._pts_num

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
    def add_points(self, pts, geo_feats=None, col_feats=None):
        """append points [n,3] (+ features, default N(0,0.1) as neural_point.py:243-246) and
        rebuild the search structure"""
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

    def retrain_updated_points(self):
        """after positions changed (deformation): rebuild (neural_point.py:441-444)"""
        self.index.set_points(self._cloud_pos)

    # ---- search -----------------------------------------------------------------------
    def find_neighbors_faiss(self, pos, step='add', retrain=False, is_pts_grad=False, dynamic_radius=None):
        """neural_point.py:264-313 -> (D [Q,nn] squared distances, I [Q,nn] int64, neighbor_num [Q] int32)"""
        assert step in ['add', 'query']
        if retrain:
            self.index.set_points(self._cloud_pos)
        if step == 'query':
            radius = self.radius_query
        else:
            radius = self.radius_add if not is_pts_grad else self.radius_min
        if dynamic_radius is not None:
            assert pos.shape[0] == dynamic_radius.shape[0], 'shape mis-match for input points and dynamic radius'
        return self.index.search(pos, self.nn_num, radius=radius, radius_per_query=dynamic_radius)

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
        _, _, nn_num = self.find_neighbors_faiss(pts, step='query')
        occ = nn_num.reshape(n_rays, intervals) > 0
        invalid = occ.sum(-1) < 2
        z_section = np.linspace(near, far, intervals)
        z_total = np.tile(np.linspace(near, far, num), (n_rays, 1))
        occ_np = occ.cpu().numpy()
        inv_np = invalid.cpu().numpy()
        for r in np.nonzero(~inv_np)[0]:
            c = np.nonzero(occ_np[r])[0]
            z_total[r] = np.linspace(z_section[c[0]], z_section[c[1]], num=num)
        return torch.from_numpy(z_total).float().to(self.device), invalid
