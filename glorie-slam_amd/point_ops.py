"""Torch-facing wrappers of the renderer entry points of libglorie_hip (KNN cell list,
IDW feature gather, fused decoder, compositing)."""
import torch

from . import _lib as L

KNN_GRID_BYTES = 64


class KnnIndex:
    """Device-resident cell list over a point set; stands in for the faiss IndexIVFFlat of
    the reference (neural_point.py:56-60).  `add`/`reset`/`train` mirror the faiss calls the
    reference makes: the structure is simply rebuilt (a few launches, no host sync)."""

    def __init__(self, device, cell_size=0.08, max_cells=1 << 21, ctx=None):
        self.device = torch.device(device)
        self.cell_size = float(cell_size)
        self.max_cells = int(max_cells)
        self.ctx = ctx
        self.points = None
        self.ntotal = 0
        self.is_trained = True
        self.cell_start = torch.zeros(self.max_cells + 1, dtype=torch.int32, device=self.device)
        self.grid = torch.zeros(KNN_GRID_BYTES // 4, dtype=torch.int32, device=self.device)
        self.sorted_pos = torch.zeros(0, 4, dtype=torch.float32, device=self.device)

    def _ctx(self):
        if self.ctx is None:
            self.ctx = L.default_context()
        return self.ctx

    def train(self, xb):
        self.is_trained = True

    def reset(self):
        self.points = None
        self.ntotal = 0
        self._rebuild()

    def add(self, xb):
        xb = xb.detach().to(self.device, torch.float32).reshape(-1, 3)
        self.points = xb.clone() if self.points is None else torch.cat([self.points, xb], 0)
        self.ntotal = self.points.shape[0]
        self._rebuild()

    def set_points(self, pts):
        self.points = pts.detach().to(self.device, torch.float32).reshape(-1, 3).contiguous()
        self.ntotal = self.points.shape[0]
        self._rebuild()

    def _rebuild(self):
        n = self.ntotal
        pts = self.points.contiguous() if n else None
        self.sorted_pos = torch.empty(max(n, 1), 4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.load().glorie_knn_build(self._ctx().handle, L.ptr(pts), n, self.cell_size,
                                              self.max_cells, L.ptr(self.sorted_pos),
                                              L.ptr(self.cell_start), L.ptr(self.grid),
                                              L.stream_ptr()), "glorie_knn_build")

    def search(self, q, k=8, radius=0.0, radius_per_query=None):
        """-> D [Q,k] f32, I [Q,k] int64, neighbor_num [Q] int32 (count of D < r^2)"""
        q = q.detach().to(self.device, torch.float32).reshape(-1, 3).contiguous()
        Q = q.shape[0]
        D = torch.empty(Q, k, dtype=torch.float32, device=self.device)
        I = torch.empty(Q, k, dtype=torch.int64, device=self.device)
        nn = torch.empty(Q, dtype=torch.int32, device=self.device)
        rp = None
        if radius_per_query is not None:
            rp = radius_per_query.detach().to(self.device, torch.float32).reshape(-1).contiguous()
            if rp.shape[0] != Q:
                raise RuntimeError("shape mis-match for input points and dynamic radius")
        with torch.cuda.device(self.device):
            L.check(L.load().glorie_knn_query(L.ptr(self.sorted_pos), L.ptr(self.cell_start),
                                              L.ptr(self.grid), L.ptr(q), Q, k, float(radius), L.ptr(rp),
                                              L.ptr(D), L.ptr(I), L.ptr(nn), L.stream_ptr()),
                    "glorie_knn_query")
        return D, I, nn


def idw_gather(D, I, nn, feats, radius=0.0, radius_per_query=None, min_nn=2, expo=False,
               return_weights=False):
    """decoder.py:130-173: c [Q,32], has_neighbors [Q] bool (, weights [Q,8])"""
    L.need_cuda(D, I, nn, feats)
    Q, k = D.shape
    feats = feats.contiguous()
    c = torch.empty(Q, feats.shape[1], dtype=torch.float32, device=D.device)
    has = torch.empty(Q, dtype=torch.uint8, device=D.device)
    w = torch.empty(Q, k, dtype=torch.float32, device=D.device) if return_weights else None
    rp = radius_per_query.reshape(-1).contiguous().float() if radius_per_query is not None else None
    L.check(L.load().glorie_idw_gather(L.ptr(D.contiguous()), L.ptr(I.contiguous()), L.ptr(nn.contiguous()),
                                       L.ptr(feats), Q, k, feats.shape[1], float(radius), L.ptr(rp),
                                       int(min_nn), int(bool(expo)), L.ptr(c), L.ptr(w), L.ptr(has),
                                       L.stream_ptr()), "glorie_idw_gather")
    return (c, has.bool(), w) if return_weights else (c, has.bool())


def composite(raw, z_vals, coef=0.1, return_weights=True):
    """raw2outputs_nerf_color (common.py:261-299): raw [R,S,4], z_vals [R,S]
    -> depth [R], var [R], rgb [R,3], weights [R,S]"""
    L.need_cuda(raw, z_vals)
    raw = raw.contiguous().float()
    z_vals = z_vals.contiguous().float()
    R, S, _ = raw.shape
    dev = raw.device
    depth = torch.empty(R, device=dev)
    var = torch.empty(R, device=dev)
    rgb = torch.empty(R, 3, device=dev)
    w = torch.empty(R, S, device=dev) if return_weights else None
    L.check(L.load().glorie_composite(L.ptr(raw), L.ptr(z_vals), R, S, float(coef), L.ptr(depth),
                                      L.ptr(var), L.ptr(rgb), L.ptr(w), L.stream_ptr()),
            "glorie_composite")
    return depth, var, rgb, w
