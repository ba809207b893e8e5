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

    def search(self, q, k=8, radius=0.0, radius_per_query=None, image_layout=None, weights=None):
        """-> D [Q,k] f32, I [Q,k] int64, neighbor_num [Q] int32 (count of D < r^2).
        image_layout = (samples_per_ray, image_w) when q holds the samples of row-major image rays (render_img): the same
        result from a search that walks the image in 16 x 16 pixel patches (glorie_knn_query_image).
        weights = (min_nn, expo[, ball_only]): also return the IDW weights [Q,8] and the neighbour mask [Q] uint8 of
        idw_gather, written by the search launch itself (k == 8; glorie_knn_query_weights) -> (D, I, nn, w, has).
        ball_only: the search stops at the query radius (slots beyond it - weight 0 - are then not the exact k-NN)."""
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
        if weights is not None:
            if k != 8:
                raise RuntimeError("KnnIndex.search: weights need k == 8")
            w = torch.empty(Q, 8, dtype=torch.float32, device=self.device)
            has = torch.empty(Q, dtype=torch.uint8, device=self.device)
            S, image_w = (int(image_layout[0]), int(image_layout[1])) if image_layout is not None else (1, 0)
            with torch.cuda.device(self.device):
                L.check(L.load().glorie_knn_query_weights(L.ptr(self.sorted_pos), L.ptr(self.cell_start), L.ptr(self.grid),
                                                          L.ptr(q), Q, float(radius), L.ptr(rp), L.ptr(D), L.ptr(I), L.ptr(nn),
                                                          S, image_w, int(weights[0]), int(bool(weights[1])),
                                                          int(bool(weights[2])) if len(weights) > 2 else 0, L.ptr(w),
                                                          L.ptr(has), L.stream_ptr()), "glorie_knn_query_weights")
            return D, I, nn, w, has
        with torch.cuda.device(self.device):
            if image_layout is not None:
                S, image_w = int(image_layout[0]), int(image_layout[1])
                L.check(L.load().glorie_knn_query_image(L.ptr(self.sorted_pos), L.ptr(self.cell_start),
                                                        L.ptr(self.grid), L.ptr(q), Q, k, float(radius), L.ptr(rp),
                                                        L.ptr(D), L.ptr(I), L.ptr(nn), S, image_w, L.stream_ptr()),
                        "glorie_knn_query_image")
            else:
                L.check(L.load().glorie_knn_query(L.ptr(self.sorted_pos), L.ptr(self.cell_start),
                                                  L.ptr(self.grid), L.ptr(q), Q, k, float(radius), L.ptr(rp),
                                                  L.ptr(D), L.ptr(I), L.ptr(nn), L.stream_ptr()),
                        "glorie_knn_query")
        return D, I, nn


def idw_gather(D, I, nn, feats, radius=0.0, radius_per_query=None, min_nn=2, expo=False,
               return_weights=False, raw_mask=False):
    """decoder.py:130-173: c [Q,32], has_neighbors [Q] bool (, weights [Q,8])"""
    L.need_cuda(D, I, nn)
    Q, k = D.shape
    weights_only = feats is None                      # weights + mask only (the geometry kernel interpolates)
    feats = feats.contiguous() if feats is not None else None
    c = torch.empty(Q, feats.shape[1], dtype=torch.float32, device=D.device) if not weights_only else None
    has = torch.empty(Q, dtype=torch.uint8, device=D.device)
    w = torch.empty(Q, k, dtype=torch.float32, device=D.device) if return_weights else None
    rp = radius_per_query.reshape(-1).contiguous().float() if radius_per_query is not None else None
    L.check(L.load().glorie_idw_gather(L.ptr(D.contiguous()), L.ptr(I.contiguous()), L.ptr(nn.contiguous()),
                                       L.ptr(feats), Q, k, 32 if weights_only else feats.shape[1], float(radius), L.ptr(rp),
                                       int(min_nn), int(bool(expo)), L.ptr(c), L.ptr(w), L.ptr(has),
                                       L.stream_ptr()), "glorie_idw_gather")
    if not raw_mask:
        has = has.bool()
    return (c, has, w) if return_weights else (c, has)


def idw_gather2(D, I, nn, feats_a, feats_b, radius=0.0, radius_per_query=None, min_nn=2, expo=False):
    """idw_gather on the geometry and the colour table in one launch -> c_a [Q,32], c_b [Q,32], has [Q] bool, w [Q,8]"""
    L.need_cuda(D, I, nn, feats_a, feats_b)
    Q, k = D.shape
    ca = torch.empty(Q, 32, dtype=torch.float32, device=D.device)
    cb = torch.empty(Q, 32, dtype=torch.float32, device=D.device)
    has = torch.empty(Q, dtype=torch.uint8, device=D.device)
    w = torch.empty(Q, k, dtype=torch.float32, device=D.device)
    rp = radius_per_query.reshape(-1).contiguous().float() if radius_per_query is not None else None
    L.check(L.load().glorie_idw_gather2(L.ptr(D.contiguous()), L.ptr(I.contiguous()), L.ptr(nn.contiguous()),
                                        L.ptr(feats_a.contiguous()), L.ptr(feats_b.contiguous()), Q, k, feats_a.shape[1],
                                        float(radius), L.ptr(rp), int(min_nn), int(bool(expo)), L.ptr(ca), L.ptr(cb),
                                        L.ptr(w), L.ptr(has), L.stream_ptr()), "glorie_idw_gather2")
    return ca, cb, has.bool(), w


_T_LIN = {}


def t_lin(dev, S):
    """the [S] sample fractions of a ray, one per (device, S); created on the CALLER's stream - Renderer.prepare_frame
    calls this before the frame's batches fork onto their own streams"""
    key = (str(dev), int(S))
    if key not in _T_LIN:
        _T_LIN[key] = torch.linspace(0.0, 1.0, steps=S, device=dev)
    return _T_LIN[key]


def ray_samples(rays_o, rays_d, depth, radius, S, near_s, far_s):
    """Renderer.py:106-125,177-184 for rays with a depth prior -> z_vals [R,S], pts [R*S,3],
    views [R*S,3], radius per sample [R*S] (None without `radius`), n_zero (device int32 [1]: rays with
    depth <= 0, which need the general path)."""
    L.need_cuda(rays_o, rays_d, depth)
    dev = rays_o.device
    R = rays_o.shape[0]
    tl = t_lin(dev, S)
    f = lambda t: t.detach().reshape(-1).contiguous().float()
    z = torch.empty(R, S, device=dev)
    pts = torch.empty(R * S, 3, device=dev)
    views = torch.empty(R * S, 3, device=dev)
    rs = torch.empty(R * S, device=dev) if radius is not None else None
    nz = torch.zeros(1, dtype=torch.int32, device=dev)
    o, d, g = f(rays_o), f(rays_d), f(depth)
    r = f(radius) if radius is not None else None
    if g.shape[0] != R or (r is not None and r.shape[0] != R):
        raise RuntimeError("ray_samples: depth / radius must have one entry per ray")
    L.check(L.load().glorie_ray_samples(L.ptr(o), L.ptr(d), L.ptr(g), L.ptr(r), L.ptr(tl), R, int(S),
                                        float(near_s), float(far_s), L.ptr(z), L.ptr(pts), L.ptr(views),
                                        L.ptr(rs), L.ptr(nz), L.stream_ptr()), "glorie_ray_samples")
    return z, pts, views, rs, nz


def camera_block(c2w, fx, fy, cx, cy, device):
    """the 16 floats glorie_ray_samples_camera reads: c2w rows 0..2, 1/fx, 1/fy (fp32 reciprocals), cx, cy"""
    import numpy as np
    c = torch.as_tensor(c2w, dtype=torch.float32).to(device)[:3, :4].reshape(-1)
    # torch evaluates `tensor / python_scalar` as tensor * float32(1.0 / scalar), the reciprocal taken in double
    k = np.array([np.float32(1.0 / float(fx)), np.float32(1.0 / float(fy)), cx, cy], dtype=np.float32)
    return torch.cat([c, torch.from_numpy(k).to(device)]).contiguous()


def ray_samples_camera(cam, image_w, first_pixel, depth, radius, S, near_s, far_s):
    """ray_samples for R = len(depth) consecutive row-major pixels of the view `cam` (camera_block) starting at
    first_pixel: get_rays (common.py:302-322) happens inside the kernel"""
    L.need_cuda(cam, depth)
    dev = depth.device
    f = lambda t: t.detach().reshape(-1).contiguous().float()
    g = f(depth)
    R = g.shape[0]
    tl = t_lin(dev, S)
    z = torch.empty(R, S, device=dev)
    pts = torch.empty(R * S, 3, device=dev)
    views = torch.empty(R * S, 3, device=dev)
    rs = torch.empty(R * S, device=dev) if radius is not None else None
    nz = torch.zeros(1, dtype=torch.int32, device=dev)
    r = f(radius) if radius is not None else None
    if r is not None and r.shape[0] != R:
        raise RuntimeError("ray_samples_camera: radius must have one entry per ray")
    L.check(L.load().glorie_ray_samples_camera(L.ptr(cam), int(image_w), int(first_pixel), L.ptr(g), L.ptr(r),
                                               L.ptr(tl), R, int(S), float(near_s), float(far_s), L.ptr(z),
                                               L.ptr(pts), L.ptr(views), L.ptr(rs), L.ptr(nz), L.stream_ptr()),
            "glorie_ray_samples_camera")
    return z, pts, views, rs, nz


def ray_counts(has, S, min_samples=3):
    """decoder.py:202-204: has [R*S] (bool / uint8) -> counts [R] int64, valid [R] bool"""
    L.need_cuda(has)
    h8 = has if has.dtype == torch.uint8 else has.to(torch.uint8)
    h8 = h8.contiguous()
    R = h8.numel() // S
    counts = torch.empty(R, dtype=torch.int64, device=has.device)
    valid = torch.empty(R, dtype=torch.bool, device=has.device)
    L.check(L.load().glorie_ray_counts(L.ptr(h8), R, int(S), int(min_samples), L.ptr(counts), L.ptr(valid),
                                       L.stream_ptr()), "glorie_ray_counts")
    return counts, valid


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


# --------------------------------------------------------------------------------------
# fused decoders (csrc/mlp.hip)
# --------------------------------------------------------------------------------------
def _t(w):
    """nn.Linear weight [out,in] -> K-major [in,out] fp32"""
    return w.detach().float().t().contiguous()


def _pad_rows(m, rows):
    out = torch.zeros(rows, m.shape[1], dtype=torch.float32, device=m.device)
    out[:m.shape[0]] = m
    return out


def _pad_cols(m, cols):
    out = torch.zeros(m.shape[0], cols, dtype=torch.float32, device=m.device)
    out[:, :m.shape[1]] = m
    return out


def pack_decoders(decoders):
    """Flatten a POINT module into the parameter buffer of glorie_render_mlp (layout mirrored in
    csrc/mlp.hip: GeoParams | NbParams | ColParams)."""
    g, c = decoders.geo_decoder, decoders.color_decoder
    dev = next(decoders.parameters()).device
    f = lambda t: t.detach().float().reshape(-1).to(dev)
    parts = []
    # geometry
    parts.append(f(_pad_cols(g.embedder._B.detach().float().to(dev), 96)))
    parts.append(f(_pad_rows(_t(g.pts_linears[0].weight), 96)))
    parts.append(f(_t(g.pts_linears[1].weight)))
    parts.append(f(_t(g.pts_linears[2].weight)))
    w3 = _t(g.pts_linears[3].weight)                       # [125, 32]: rows 0..92 embedding, 93..124 hidden
    parts.append(f(_pad_rows(w3[:93], 96)))
    parts.append(f(w3[93:]))
    parts.append(f(_t(g.pts_linears[4].weight)))
    parts.append(f(_pad_cols(_t(g.output_linear.weight), 16)))
    parts.append(torch.cat([f(_t(l.weight)) for l in g.fc_c]))
    parts.append(torch.cat([f(l.bias) for l in g.pts_linears]))
    parts.append(torch.cat([f(l.bias) for l in g.fc_c]))
    parts.append(torch.cat([f(g.output_linear.bias), torch.zeros(3, device=dev)]))
    # per-neighbour F_theta (of the colour decoder)
    n = c.mlp_col_neighbor
    parts.append(f(c.embedder_rel_pos._B))
    parts.append(torch.zeros(2, device=dev))
    parts.append(f(_t(n.linear1.weight)))
    parts.append(f(n.linear1.bias))
    parts.append(f(_t(n.linear2.weight)))
    parts.append(f(n.linear2.bias))
    # colour
    parts.append(f(c.embedder._B.to(dev)))
    parts.append(f(c.embedder_view_direction._B.to(dev)))
    parts.append(f(_t(c.pts_linears[0].weight)))
    parts.append(f(_t(c.pts_linears[1].weight)))
    parts.append(f(_t(c.pts_linears[2].weight)))
    w3 = _t(c.pts_linears[3].weight)                       # [208, 128]: rows 0..79 embedding
    parts.append(f(w3[:80]))
    parts.append(f(w3[80:]))
    parts.append(f(_t(c.pts_linears[4].weight)))
    parts.append(f(_pad_cols(_t(c.output_linear.weight), 16)))
    parts.append(torch.cat([f(_t(l.weight)) for l in c.fc_c]))
    parts.append(torch.cat([f(l.bias) for l in c.pts_linears]))
    parts.append(torch.cat([f(l.bias) for l in c.fc_c]))
    parts.append(torch.cat([f(c.output_linear.bias), torch.zeros(1, device=dev)]))
    # colour weights once more, in the 32-row chunk order the colour kernel streams through LDS:
    # W0(80->96) Fc0 | W1 Fc1 | W2 Fc2 | W3e(80->96) W3h Fc3 | W4 Fc4   = 27 chunks of [32,128];
    # inside a row, output column 16*to + r sits at (to >> 2) * 64 + 4*r + (to & 3) (two 16-byte LDS
    # reads per lane fetch the eight MFMA A operands of a k-step)
    w3 = _t(c.pts_linears[3].weight)
    fc = [_t(l.weight) for l in c.fc_c]
    seq = [_pad_rows(_t(c.pts_linears[0].weight), 96), fc[0], _t(c.pts_linears[1].weight), fc[1],
           _t(c.pts_linears[2].weight), fc[2], _pad_rows(w3[:80], 96), w3[80:], fc[3],
           _t(c.pts_linears[4].weight), fc[4]]
    chunks = torch.cat([m.to(dev) for m in seq], 0)
    assert chunks.shape == (27 * 32, 128), chunks.shape
    chunks16 = chunks.detach().float()
    chunks = chunks.reshape(-1, 2, 4, 16).permute(0, 1, 3, 2).reshape(-1, 128)
    parts.append(f(chunks))
    # geometry weights once more, as the LDS image of the geometry kernel: 480 K-rows
    # W0(93->96) Fc0 | W1 Fc1 | W2 Fc2 | W3e(93->96) W3h Fc3 | W4 Fc4, output 16*to + r of a row at 2*r + to,
    # followed by the output layer [32,16] in natural order
    w3 = _t(g.pts_linears[3].weight)
    fc = [_t(l.weight) for l in g.fc_c]
    seq = [_pad_rows(_t(g.pts_linears[0].weight), 96), fc[0], _t(g.pts_linears[1].weight), fc[1],
           _t(g.pts_linears[2].weight), fc[2], _pad_rows(w3[:93], 96), w3[93:], fc[3],
           _t(g.pts_linears[4].weight), fc[4]]
    rows = torch.cat([m.to(dev) for m in seq], 0)
    assert rows.shape == (480, 32), rows.shape
    geo_rows16 = rows.detach().float()
    parts.append(f(rows.reshape(480, 2, 16).permute(0, 2, 1)))
    parts.append(f(_pad_cols(_t(g.output_linear.weight), 16)))
    # the colour chunks a third time, as fp16 MFMA A fragments of the 3-term split (mlp_col_v4_kernel):
    # [chunk][hi|lo][out block to][lane = 16 g + i][slot s] halfs, chunk row 16 (s >> 2) + 4 g + (s & 3), output 16 to + i;
    # hi = fp16(w), lo = fp16(w - hi); carried as raw bits in the float buffer (4096 floats per chunk)
    t6 = chunks16.reshape(27, 2, 4, 4, 8, 16).permute(0, 4, 2, 5, 1, 3)       # [chunk][to][g][i][s >> 2][s & 3]
    hi = t6.half()
    lo = (t6 - hi.float()).half()
    frag = torch.stack([hi, lo], 1).reshape(27, -1).contiguous()              # [chunk][2 * 8 * 64 * 8]
    parts.append(frag.view(torch.float32).reshape(-1))
    # per-neighbour W1 [52, 128] the same way (mlp_nb_v4_kernel): chunk 0 = the 20 embedding rows (slot s < 5 <-> row 4 s + g,
    # slots 5..7 zero), chunk 1 = the 32 colour-feature rows (slot s <-> row 20 + 16 (s >> 2) + 4 g + (s & 3))
    w1 = _t(n.linear1.weight).detach().float().to(dev)                        # [52, 128]
    c0 = torch.zeros(8, 4, 128, device=dev)                                   # [s][g][out]
    c0[:5] = w1[:20].reshape(5, 4, 128)
    c1 = w1[20:52].reshape(2, 4, 4, 128).permute(0, 2, 1, 3).reshape(8, 4, 128)   # [sh][g][sl] -> [s = 4 sh + sl][g]
    both = torch.stack([c0, c1], 0).reshape(2, 8, 4, 8, 16).permute(0, 3, 2, 4, 1)  # [chunk][to][g][i][s]
    hi = both.half()
    lo = (both - hi.float()).half()
    frag = torch.stack([hi, lo], 1).reshape(-1).contiguous()                  # [chunk][hi|lo][to][g][i][s]
    parts.append(frag.view(torch.float32).reshape(-1))
    # ... and the geometry K-rows (mlp_geo_v4_kernel): 15 chunks, 2 output blocks, 1024 floats per chunk
    t6 = geo_rows16.reshape(15, 2, 4, 4, 2, 16).permute(0, 4, 2, 5, 1, 3)     # [chunk][to][g][i][s >> 2][s & 3]
    hi = t6.half()
    lo = (t6 - hi.float()).half()
    frag = torch.stack([hi, lo], 1).reshape(15, -1).contiguous()              # [chunk][2 * 2 * 64 * 8]
    parts.append(frag.view(torch.float32).reshape(-1))
    packed = torch.cat(parts).contiguous()
    expect = int(L.load().glorie_decoder_pack_floats())
    if packed.numel() != expect:
        raise L.GlorieError(f"decoder pack has {packed.numel()} floats, library expects {expect}")
    return packed


def render_mlp(packed, pts, views, cloud_pos, col_feats, c_geo, I, weights, has, stage="color", geo_feats=None,
               precise=False, range_flag=None, gather_phase_only=False):
    """raw [Q,4] = (rgb, occ) for samples `pts` given their neighbours; occ = -100 where
    has == False.  stage 'geometry' leaves rgb = 0.  c_geo: the interpolated geometry feature [Q,32], or None
    with geo_feats [Np,32]: the geometry kernel interpolates it itself from (I, weights).
    precise: exact-fp32 MFMA kernels instead of the 3-term fp16 split.  range_flag: device int32 [1] (zero): the split
    kernels set it when an operand left the fp16 range - the caller then repeats the call with precise=True
    (`range_checked` does both).  gather_phase_only (bench.py): the measurement instantiations of the geometry / per-neighbour
    kernels that pull the neighbour rows and drop the networks (stage_flags & 4, include/glorie_hip.h); the result is NOT a
    rendering."""
    L.need_cuda(packed, pts, c_geo if c_geo is not None else geo_feats)
    Q = pts.shape[0]
    dev = pts.device
    color = stage == "color"
    # the colour kernel writes rgb of every sample, the geometry kernel the occupancy: only the geometry
    # stage needs the zero fill
    raw = (torch.empty if color else torch.zeros)(Q, 4, dtype=torch.float32, device=dev)
    scratch = torch.empty(Q, 32, dtype=torch.float32, device=dev) if color else None
    has8 = (has if has.dtype == torch.uint8 else has.to(torch.uint8)).contiguous()
    L.check(L.load().glorie_render_mlp(
        L.ptr(packed), L.ptr(pts.contiguous().float()),
        L.ptr(views.contiguous().float()) if color else None,
        L.ptr(cloud_pos.contiguous()) if color else None, L.ptr(col_feats.contiguous()) if color else None,
        L.ptr(c_geo.contiguous()) if c_geo is not None else None,
        L.ptr(geo_feats.contiguous()) if c_geo is None else None,
        L.ptr(I.contiguous()) if (color or c_geo is None) else None,
        L.ptr(weights.contiguous()) if (color or c_geo is None) else None, L.ptr(has8), Q, L.ptr(scratch),
        L.ptr(raw), int(color) | (2 if precise else 0) | (4 if gather_phase_only else 0), L.ptr(range_flag),
        L.stream_ptr()), "glorie_render_mlp")
    return raw


class RangeGuard:
    """the device word the split decoder kernels raise when an operand does not fit fp16 (|x| > 65504 or NaN), and the
    host side of the protocol: `tripped()` reads and clears it (one host synchronisation)"""

    def __init__(self, device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)

    def tripped(self):
        bad = int(self.flag.item()) != 0
        if bad:
            self.flag.zero_()
        return bad
