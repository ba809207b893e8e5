"""Drop-in for the reference's pybind11 extension ``droid_backends``
(/root/reference/src/lib/droid.cpp:239-252) on top of the C ABI of libglorie_hip.so.

Same function names, argument order, return shapes and error behaviour (non-contiguous
input -> RuntimeError, like CHECK_CONTIGUOUS at droid.cpp:85-86).  ``ba`` mutates ``poses``
and ``disps`` in place exactly like ``ba_cuda``.  The backward ops and ``projmap`` are not on
the inference hot path and raise NotImplementedError.

Extra entry points (fused forms the reference builds out of several torch ops):
``corr_lookup_pyramid``, ``reproject``, ``cvx_upsample``.
"""
import ctypes
import os

import torch

from . import _lib as L


_CHECK_STATUS = os.environ.get("GLORIE_CHECK_STATUS", "0") not in ("", "0")


def _i64(t, name):
    if t.dtype != torch.int64:
        raise RuntimeError(f"{name} must be int64 (the reference kernels read `long`)")
    return t


# --------------------------------------------------------------------------------------
# correlation
# --------------------------------------------------------------------------------------
def corr_index_forward(volume, coords, radius):
    """volume [N,h1,w1,h2,w2] (f16|f32), coords [N,2,h1,w1] f32 -> [corr [N,2r+1,2r+1,h1,w1]]
    reference: droid.cpp:172-180, correlation_kernels.cu:126-155"""
    L.need_cuda(volume, coords)
    L.need_contiguous(volume=volume, coords=coords)
    if coords.dtype != torch.float32:
        raise RuntimeError("coords must be float32")
    N, h1, w1, h2, w2 = volume.shape
    rd = 2 * radius + 1
    out = torch.empty((N, rd, rd, h1, w1), dtype=volume.dtype, device=volume.device)
    lib = L.load()
    L.check(lib.glorie_corr_index_fwd(L.ptr(volume), L.ptr(coords), L.ptr(out), N, h1, w1, h2, w2,
                                      radius, L.dtype_code(volume), L.stream_ptr()),
            "glorie_corr_index_fwd")
    return [out]


def corr_lookup_pyramid(pyramid, coords, radius):
    """Fused CorrBlock.__call__ body (corr.py:43-53): pyramid = list of level tensors
    [N,h1,w1,h2>>l,w2>>l]; coords [N,2,h1,w1] f32 UNscaled -> [N, L*(2r+1)^2, h1, w1]."""
    L.need_cuda(coords, *pyramid)
    L.need_contiguous(coords=coords, **{f"level{i}": v for i, v in enumerate(pyramid)})
    N, h1, w1, h2, w2 = pyramid[0].shape
    nl = len(pyramid)
    for l, v in enumerate(pyramid):
        if tuple(v.shape) != (N, h1, w1, h2 >> l, w2 >> l) or v.dtype != pyramid[0].dtype:
            raise RuntimeError(f"pyramid level {l} has shape {tuple(v.shape)}")
    rd = 2 * radius + 1
    out = torch.empty((N, nl * rd * rd, h1, w1), dtype=pyramid[0].dtype, device=coords.device)
    arr = (ctypes.c_void_p * nl)(*[v.data_ptr() for v in pyramid])
    lib = L.load()
    L.check(lib.glorie_corr_lookup_pyramid(ctypes.cast(arr, ctypes.c_void_p), nl, L.ptr(coords),
                                           L.ptr(out), N, h1, w1, h2, w2, radius,
                                           L.dtype_code(pyramid[0]), L.stream_ptr()),
            "glorie_corr_lookup_pyramid")
    return out


def tile_corr_level(vol):
    """[P, h2, w2] fp16 plane stack -> tiled [P, ceil(h2/4)*ceil(w2/8)*32] (64-byte blocks of 4 rows x
    8 columns, zero padded), allocated with one spare plane on either side (read slack of the
    unaligned block-row loads of the tiled lookup kernel)"""
    P, h2, w2 = vol.shape
    nby, nbx = (h2 + 3) // 4, (w2 + 7) // 8
    store = torch.zeros((P + 2, nby, nbx, 4, 8), dtype=vol.dtype, device=vol.device)
    padded = torch.nn.functional.pad(vol, (0, nbx * 8 - w2, 0, nby * 4 - h2))
    store[1:P + 1] = padded.view(P, nby, 4, nbx, 8).permute(0, 1, 3, 2, 4)
    return store.view(P + 2, -1)[1:P + 1]


def corr_lookup_pyramid_tiled(pyramid, coords, h2, w2):
    """CorrBlock.__call__ body on tiled levels (see tile_corr_level): pyramid = list of
    [N*h1*w1, plane_l] fp16, coords [N,2,h1,w1] f32 UNscaled -> [N, L*49, h1, w1]; bit-identical to
    corr_lookup_pyramid on the row-major volumes."""
    L.need_cuda(coords, *pyramid)
    N, _, h1, w1 = coords.shape
    nl = len(pyramid)
    for l, v in enumerate(pyramid):
        nby, nbx = ((h2 >> l) + 3) // 4, ((w2 >> l) + 7) // 8
        if v.dtype != torch.float16 or v.dim() != 2 or v.shape[0] != N * h1 * w1 or v.shape[1] != nby * nbx * 32 \
                or v.stride(1) != 1 or v.stride(0) != v.shape[1]:
            raise RuntimeError(f"tiled pyramid level {l} has shape {tuple(v.shape)}")
    if not coords.is_contiguous() or coords.dtype != torch.float32:
        raise RuntimeError("coords must be contiguous float32")
    out = torch.empty((N, nl * 49, h1, w1), dtype=torch.float16, device=coords.device)
    arr = (ctypes.c_void_p * nl)(*[v.data_ptr() for v in pyramid])
    L.check(L.load().glorie_corr_lookup_pyramid_tiled(ctypes.cast(arr, ctypes.c_void_p), nl, L.ptr(coords),
                                                      L.ptr(out), N, h1, w1, h2, w2, L.stream_ptr()),
            "glorie_corr_lookup_pyramid_tiled")
    return out


def corr_lookup_tiled_cl(pyramid, coords, h2, w2, slots=None, interleaved=False):
    """the tiled lookup with a channels-last result (glorie_corr_lookup_tiled_cl): pyramid = 4 tiled levels (stacked in
    edge order, or arena tensors with `slots` int32 [N]), coords f32 UNscaled, planar [N,2,h1,w1] or - interleaved=True -
    [N,h1,w1,2] (the reprojection's layout: no permute + copy) -> fp16 [N,256,h1,w1] in channels_last memory format,
    channel l*64 + dy*8 + dx = the reference's channel l*49 + dx*7 + dy, padding zero"""
    L.need_cuda(coords, *pyramid)
    if coords.dim() != 4 or coords.shape[3 if interleaved else 1] != 2:
        raise RuntimeError("coords must be [N,2,h1,w1] (planar) or [N,h1,w1,2] (interleaved)")
    if interleaved:
        N, h1, w1, _ = coords.shape
    else:
        N, _, h1, w1 = coords.shape
    if len(pyramid) != 4:
        raise RuntimeError("corr_lookup_tiled_cl: 4 levels expected")
    if not coords.is_contiguous() or coords.dtype != torch.float32:
        raise RuntimeError("coords must be contiguous float32")
    out = torch.empty((N, 256, h1, w1), dtype=torch.float16, device=coords.device, memory_format=torch.channels_last)
    arr = (ctypes.c_void_p * 4)(*[v.data_ptr() for v in pyramid])
    L.check(L.load().glorie_corr_lookup_tiled_cl(ctypes.cast(arr, ctypes.c_void_p), 4, L.ptr(slots), L.ptr(coords),
                                                 int(bool(interleaved)), L.ptr(out), N, h1, w1, h2, w2, L.stream_ptr()),
            "glorie_corr_lookup_tiled_cl")
    return out


# ---- displacement-major, source-tiled pyramid (csrc/corr_dm.hip) ---------------------------------------------
def dm_shape(h, w, l):
    """(ntiles, h>>l, Wp) of level l, Wp = (w>>l) rounded up to even: levels are [slots][ntiles][(h>>l) * Wp][64] fp16 -
    per (tile, dy) the line of displacement-column PAIR dx2 holds, per lane, the two halfs dx = 2 dx2 and 2 dx2 + 1"""
    return ((h + 7) // 8) * ((w + 7) // 8), h >> l, ((w >> l) + 1) & ~1


def _dm_index(n_src, n_dst, l, device, n_mod=None):
    """[source coordinate s][displacement index d] -> target coordinate t, i.e. the inverse of
    d = (t - (s >> l) + (n_dst >> 1)) mod n_mod (include/glorie_hip.h: glorie_corr_dm_build); n_mod = n_dst for rows, the even
    padded width for columns (t >= n_dst then names the padding column)"""
    n_mod = n_dst if n_mod is None else n_mod
    s = torch.arange(n_src, device=device)[:, None] >> l
    d = torch.arange(n_mod, device=device)[None, :]
    return (d - (n_dst >> 1) + s) % n_mod


def dm_corr_level(vol, l):
    """row-major level [N, h, w, h>>l, w>>l] -> displacement-major [N, ntiles * (h>>l) * Wp * 64] (a torch
    restatement of the layout for tests and for CorrBlock pyramids built with torch; the product path builds the
    layout directly with glorie_corr_dm_build)"""
    N, h, w, hl, wl = vol.shape
    wp = (wl + 1) & ~1
    nty, ntx = (h + 7) // 8, (w + 7) // 8
    H8, W8 = nty * 8, ntx * 8
    vp = torch.zeros((N, H8, W8, hl, wp), dtype=vol.dtype, device=vol.device)      # column wl (odd widths): zeros
    vp[:, :h, :w, :, :wl] = vol
    ty = _dm_index(H8, hl, l, vol.device)            # [sy][dy] -> ty
    tx = _dm_index(W8, wl, l, vol.device, wp)        # [sx][dx] -> tx (possibly the padding column)
    sy = torch.arange(H8, device=vol.device)[:, None, None, None]
    sx = torch.arange(W8, device=vol.device)[None, :, None, None]
    d = vp[:, sy, sx, ty[:, None, :, None], tx[None, :, None, :]]              # [N, sy, sx, dy, dx]
    d = d.view(N, nty, 8, ntx, 8, hl, wp // 2, 2).permute(0, 1, 3, 5, 6, 2, 4, 7)   # [N, ty, tx, dy, dx2, ly, lx, dx & 1]
    return d.reshape(N, -1).contiguous()


def dm_to_rowmajor(dm, h, w, l):
    """inverse of dm_corr_level: [N, ntiles*(h>>l)*Wp*64] -> [N, h, w, h>>l, w>>l]"""
    N = dm.shape[0]
    hl, wl = h >> l, w >> l
    wp = (wl + 1) & ~1
    nty, ntx = (h + 7) // 8, (w + 7) // 8
    d = dm.view(N, nty, ntx, hl, wp // 2, 8, 8, 2).permute(0, 1, 5, 2, 6, 3, 4, 7).reshape(N, nty * 8, ntx * 8, hl, wp)[:, :h, :w]
    dy = (torch.arange(hl, device=dm.device)[None, :] - (torch.arange(h, device=dm.device)[:, None] >> l) + (hl >> 1)) % hl
    dx = (torch.arange(wl, device=dm.device)[None, :] - (torch.arange(w, device=dm.device)[:, None] >> l) + (wl >> 1)) % wp
    sy = torch.arange(h, device=dm.device)[:, None, None, None]
    sx = torch.arange(w, device=dm.device)[None, :, None, None]
    return d[:, sy, sx, dy[:, None, :, None], dx[None, :, None, :]].contiguous()


def corr_dm_lookup(levels, coords, h, w, slots=None, interleaved=False, want_corr=True, enc_w=None, enc_b=None,
                   enc_out=None):
    """glorie_corr_dm_lookup: levels = 4 displacement-major level tensors (stacked in edge order, or arena tensors with
    `slots` int32 [N]); coords f32 UNscaled, planar [N,2,h,w] or - interleaved - [N,h,w,2].
    want_corr: return the lookup as fp16 [N,256,h,w] channels_last (channel l*64 + j*8 + i, as corr_lookup_tiled_cl).
    enc_w / enc_b / enc_out: run corr_encoder[0] behind the lookup (update_ops.pack_corr_encoder_dm(weight), bias f32 [128],
    enc_out a channels-last fp16 [N,128,h,w] map or a 128-channel slice of a wider one)."""
    L.need_cuda(coords, *levels)
    if len(levels) != 4:
        raise RuntimeError("corr_dm_lookup: 4 levels expected")
    if coords.dim() != 4 or coords.shape[3 if interleaved else 1] != 2:
        raise RuntimeError("coords must be [N,2,h,w] (planar) or [N,h,w,2] (interleaved)")
    if not coords.is_contiguous() or coords.dtype != torch.float32:
        raise RuntimeError("coords must be contiguous float32")
    N = coords.shape[0]
    if tuple(coords.shape[1:3] if interleaved else coords.shape[2:4]) != (h, w):
        raise RuntimeError("coords do not match the map size")
    for l, v in enumerate(levels):
        nt, hl, wl = dm_shape(h, w, l)
        if v.dtype != torch.float16 or not v.is_contiguous() or v.numel() % (nt * hl * wl * 64) or \
                (slots is None and v.numel() != N * nt * hl * wl * 64):
            raise RuntimeError(f"displacement-major level {l} has {v.numel()} elements")
    out = None
    if want_corr:
        out = torch.empty((N, 256, h, w), dtype=torch.float16, device=coords.device, memory_format=torch.channels_last)
    stride = 0
    if enc_out is not None:
        if enc_w is None or enc_b is None or tuple(enc_out.shape) != (N, 128, h, w) or enc_out.dtype != torch.float16:
            raise RuntimeError("corr_dm_lookup: enc_out must be an fp16 [N,128,h,w] map, with enc_w and enc_b")
        from .update_ops import _rows
        stride = _rows(enc_out, "enc_out")
        if tuple(enc_w.shape) != (128, 224) or enc_w.dtype != torch.float16 or not enc_w.is_contiguous():
            raise RuntimeError("corr_dm_lookup: enc_w must be fp16 [128,224] (update_ops.pack_corr_encoder_dm)")
    elif not want_corr:
        raise RuntimeError("corr_dm_lookup: nothing to compute")
    arr = (ctypes.c_void_p * 4)(*[v.data_ptr() for v in levels])
    L.check(L.load().glorie_corr_dm_lookup(ctypes.cast(arr, ctypes.c_void_p), L.ptr(slots), L.ptr(coords),
                                           int(bool(interleaved)), N, h, w, L.ptr(out), L.ptr(enc_w), L.ptr(enc_b),
                                           L.ptr(enc_out), stride, L.stream_ptr()), "glorie_corr_dm_lookup")
    return out


def cl_to_planar(cl):
    """[N,256,h,w] channels-last lookup (channel l*64 + j*8 + i) -> the reference's [N,196,h,w] (channel l*49 + i*7 + j)"""
    N, _, h, w = cl.shape
    v = cl.view(N, 4, 8, 8, h, w)[:, :, :7, :7]                    # [n][l][j][i][y][x]
    return v.permute(0, 1, 3, 2, 4, 5).reshape(N, 196, h, w).contiguous()


def corr_index_backward(volume, coords, corr_grad, radius):
    raise NotImplementedError("corr_index_backward: training-only; the SLAM hot path runs under "
                              "no_grad (factor_graph.py:213)")


def altcorr_forward(fmap1, fmap2, coords, radius):
    """fmap1 [B,H,W,C], fmap2 [B,H2,W2,C], coords [B,S,H,W,2] -> [corr [B,S,(2r+1)^2,H,W]] (f32)
    reference: droid.cpp:195-205, altcorr_kernel.cu:290-319"""
    L.need_cuda(fmap1, fmap2, coords)
    L.need_contiguous(fmap1=fmap1, fmap2=fmap2, coords=coords)
    if fmap1.dtype != torch.float32 or fmap2.dtype != torch.float32:
        raise RuntimeError("altcorr_forward takes float32 feature maps (corr.py:125 casts)")
    B, H, W, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    rd = 2 * radius + 1
    out = torch.empty((B, S, rd * rd, H, W), dtype=torch.float32, device=fmap1.device)
    lib = L.load()
    L.check(lib.glorie_altcorr_fwd(L.ptr(fmap1), L.ptr(fmap2), L.ptr(coords), L.ptr(out), B, S, H,
                                   W, H2, W2, C, radius, L.stream_ptr()), "glorie_altcorr_fwd")
    return [out]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    raise NotImplementedError("altcorr_backward: training-only, not on the SLAM hot path")


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------
def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """reference: droid.cpp:122-138, droid_kernels.cu:1441-1463"""
    L.need_cuda(poses, disps, intrinsics, ii, jj)
    L.need_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, ii=ii, jj=jj)
    _i64(ii, "ii"), _i64(jj, "jj")
    K = ii.shape[0]
    h, w = disps.shape[1:]
    dist = torch.empty((K,), dtype=torch.float32, device=poses.device)
    L.check(L.load().glorie_frame_distance(L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), L.ptr(ii),
                                           L.ptr(jj), L.ptr(dist), K, h, w, float(beta),
                                           L.stream_ptr()), "glorie_frame_distance")
    return dist


def projmap(poses, disps, intrinsics, ii, jj):
    raise NotImplementedError("projmap is exported by the reference but never called "
                              "(SURVEY.md section 2.1)")


def iproj(poses, disps, intrinsics):
    """reference: droid.cpp:161-169, droid_kernels.cu:1521-1544"""
    L.need_cuda(poses, disps, intrinsics)
    L.need_contiguous(poses=poses, disps=disps, intrinsics=intrinsics)
    num, h, w = disps.shape
    pts = torch.empty((num, h, w, 3), dtype=torch.float32, device=disps.device)
    L.check(L.load().glorie_iproj(L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), L.ptr(pts), num, h,
                                  w, L.stream_ptr()), "glorie_iproj")
    return pts


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """reference: droid.cpp:208-224, droid_kernels.cu:1494-1518"""
    L.need_cuda(poses, disps, intrinsics, ix, thresh)
    L.need_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, ix=ix, thresh=thresh)
    _i64(ix, "ix")
    B, h, w = disps.shape
    num = ix.shape[0]
    count = torch.empty((num, h, w), dtype=torch.float32, device=disps.device)
    L.check(L.load().glorie_depth_filter(L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), L.ptr(ix),
                                         L.ptr(thresh), L.ptr(count), B, num, h, w, L.stream_ptr()),
            "glorie_depth_filter")
    return count


def reproject(poses, disps, intrinsics, ii, jj, return_valid=True, motion=None):
    """Fused pops.projective_transform(jacobian=False) (projective_ops.py:96-125).
    poses [B,7], disps [B,h,w], intrinsics [B,4] -> coords [N,h,w,2], valid [N,h,w,1].
    motion = (target [.., N, h, w, 2] f32, update_ops.PaddedFlow): the same launch also writes the motion features
    [coords - grid, target - coords].clamp(+-64) (factor_graph.py:219-221) into the padded fp16 map"""
    L.need_cuda(poses, disps, intrinsics, ii, jj)
    L.need_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, ii=ii, jj=jj)
    _i64(ii, "ii"), _i64(jj, "jj")
    N = ii.shape[0]
    h, w = disps.shape[1:]
    coords = torch.empty((N, h, w, 2), dtype=torch.float32, device=poses.device)
    valid = torch.empty((N, h, w, 1), dtype=torch.float32, device=poses.device) if return_valid else None
    if motion is not None:
        target, padded = motion
        L.need_cuda(target, padded.buf)
        if not target.is_contiguous() or target.dtype != torch.float32 or target.numel() != N * h * w * 2 \
                or not padded.fits(N, h, w, poses.device):
            raise RuntimeError("reproject: motion needs a contiguous float32 target [N,h,w,2] and a matching PaddedFlow")
        L.check(L.load().glorie_reproject_motion(L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), L.ptr(ii), L.ptr(jj),
                                                 L.ptr(coords), L.ptr(valid), L.ptr(target), L.ptr(padded.buf), N, h, w,
                                                 64.0, L.stream_ptr()), "glorie_reproject_motion")
        return coords, valid
    L.check(L.load().glorie_reproject(L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), L.ptr(ii),
                                      L.ptr(jj), L.ptr(coords), L.ptr(valid), N, h, w,
                                      L.stream_ptr()), "glorie_reproject")
    return coords, valid


def motion(coords1, coords0, target, limit=64.0):
    """[coords1 - coords0, target - coords1].clamp(+-limit) (factor_graph.py:219-221) as ONE
    launch.  coords1, target [..., N, h, w, 2] f32, coords0 [h, w, 2] -> [N, h, w, 4] f32"""
    L.need_cuda(coords1, coords0, target)
    L.need_contiguous(coords1=coords1, coords0=coords0, target=target)
    h, w = coords0.shape[-3], coords0.shape[-2]
    n = coords1.numel() // (h * w * 2)
    if target.numel() != coords1.numel() or coords0.numel() != h * w * 2:
        raise RuntimeError("motion: shape mismatch")
    out = torch.empty((n, h, w, 4), dtype=torch.float32, device=coords1.device)
    L.check(L.load().glorie_motion(L.ptr(coords1), L.ptr(coords0), L.ptr(target), L.ptr(out), n, h, w,
                                   float(limit), L.stream_ptr()), "glorie_motion")
    return out


def motion_padded(coords1, coords0, target, padded, limit=64.0):
    """motion() written as the zero-padded fp16 map the flow encoder's first convolution reads
    (update_ops.PaddedFlow, glorie_flow_conv7_padded): no fp32 map, no conversion in the convolution"""
    L.need_cuda(coords1, coords0, target, padded.buf)
    L.need_contiguous(coords1=coords1, coords0=coords0, target=target)
    h, w = coords0.shape[-3], coords0.shape[-2]
    n = coords1.numel() // (h * w * 2)
    if target.numel() != coords1.numel() or coords0.numel() != h * w * 2 or not padded.fits(n, h, w, coords1.device):
        raise RuntimeError("motion_padded: shape mismatch")
    L.check(L.load().glorie_motion_padded(L.ptr(coords1), L.ptr(coords0), L.ptr(target), L.ptr(padded.buf), n, h, w,
                                          float(limit), L.stream_ptr()), "glorie_motion_padded")
    return padded


def valid_depth_mask(poses, disps, intrinsics, ix, mv_thresh, visible_num):
    """two-view validity mask of frames `ix` on the map stack `disps` [B,h,w]
    (DepthVideo.update_valid_depth_mask, depth_video.py:326-361) -> bool [len(ix), h, w]"""
    L.need_cuda(poses, disps, intrinsics, ix)
    L.need_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, ix=ix)
    _i64(ix, "ix")
    B, h, w = disps.shape
    num = ix.shape[0]
    mask = torch.empty((num, h, w), dtype=torch.bool, device=disps.device)
    scratch = torch.empty(num * h * w * 4 + num * 1044 + 64, dtype=torch.uint8, device=disps.device)
    L.check(L.load().glorie_valid_depth_mask(L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), L.ptr(ix), B, num, h, w,
                                             float(mv_thresh), int(visible_num), L.ptr(mask), L.ptr(scratch),
                                             L.stream_ptr()), "glorie_valid_depth_mask")
    return mask


def dspo_prepare(poses, disps, intrinsics, mono_disps, n, mv_thresh, visible_num, mono_thres, ii, jj,
                 valid_mask, depth_scale, depth_shift, publish=None):
    """update_valid_depth_mask(up=False) + align_scale_and_shift + the mono_thres edge filter of the
    depth_scale stage (depth_video.py:228-247,326-361) in 4 launches and without a host sync.
    Writes valid_mask[:n] (bool), depth_scale[:n], depth_shift[:n]; returns (edge_on uint8 [N],
    any_on int32 [1]).  publish = (state int32 [2] on the device, address of a pinned host word): the last launch
    also stores (launch count << 1 | any_on) there (DepthVideo.await_any_on polls it)."""
    L.need_cuda(poses, disps, intrinsics, mono_disps, ii, jj, valid_mask, depth_scale, depth_shift)
    L.need_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, mono_disps=mono_disps, ii=ii, jj=jj,
                      valid_mask=valid_mask, depth_scale=depth_scale, depth_shift=depth_shift)
    _i64(ii, "ii"), _i64(jj, "jj")
    if valid_mask.dtype != torch.bool or valid_mask.shape != disps.shape:
        raise RuntimeError("valid_mask must be a bool tensor of the shape of disps")
    B, h, w = disps.shape
    N = ii.shape[0]
    dev = disps.device
    edge_on = torch.empty(N, dtype=torch.uint8, device=dev)
    any_on = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(n * h * w * 4 + n * 32 + 64, dtype=torch.uint8, device=dev)
    L.check(L.load().glorie_dspo_prepare(L.ptr(poses), L.ptr(disps), L.ptr(intrinsics), L.ptr(mono_disps), B,
                                         int(n), h, w, float(mv_thresh), int(visible_num),
                                         float(mono_thres or 0.0), L.ptr(ii), L.ptr(jj), N, L.ptr(valid_mask),
                                         L.ptr(depth_scale), L.ptr(depth_shift), L.ptr(edge_on), L.ptr(any_on),
                                         L.ptr(scratch), L.ptr(publish[0]) if publish else None,
                                         publish[1] if publish else None, L.stream_ptr()), "glorie_dspo_prepare")
    return edge_on, any_on


def cvx_upsample(disps, ix, mask, disps_up, softmax_f32=False):
    """disps_up[ix] = cvx_upsample(disps[ix], mask) in place (droid_net.py:9-23,
    depth_video.py:140-144).  mask [M,576,h,w] f16|f32."""
    L.need_cuda(disps, ix, mask, disps_up)
    _i64(ix, "ix")
    M = ix.shape[0]
    h, w = disps.shape[1:]
    if mask.numel() != M * 576 * h * w:
        raise RuntimeError(f"mask has {mask.numel()} elements, expected {M}*576*{h}*{w}")
    if mask.dim() == 4 and mask.dtype == torch.float16 and not mask.is_contiguous() \
            and mask.is_contiguous(memory_format=torch.channels_last):
        # channels-last logits straight from the upmask convolution: no layout pass
        L.need_contiguous(disps=disps, ix=ix, disps_up=disps_up)
        L.check(L.load().glorie_cvx_upsample_nhwc(L.ptr(disps), L.ptr(ix), L.ptr(mask), 576, L.ptr(disps_up),
                                                  M, h, w, int(bool(softmax_f32)), L.stream_ptr()),
                "glorie_cvx_upsample_nhwc")
        return disps_up
    L.need_contiguous(disps=disps, ix=ix, mask=mask, disps_up=disps_up)
    L.check(L.load().glorie_cvx_upsample(L.ptr(disps), L.ptr(ix), L.ptr(mask), L.ptr(disps_up), M, h,
                                         w, L.dtype_code(mask), int(bool(softmax_f32)),
                                         L.stream_ptr()), "glorie_cvx_upsample")
    return disps_up


# --------------------------------------------------------------------------------------
# bundle adjustment
# --------------------------------------------------------------------------------------
def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1,
       iterations, lm, ep, motion_only, depth_only=False, ctx=None, want_updates=True, targets_hwc=False, gate=None):
    """reference: droid.cpp:89-119, droid_kernels.cu:1314-1437.  Returns [dx, dz] of the last
    iteration (the reference's return value, unused by its caller).  poses/disps updated in
    place.  targets_hwc=True: targets / weights are [N,h,w,2] (FactorGraph's own layout) instead of the
    binding's [N,2,h,w].  gate = (flag, hits): int32 device words - the call's kernels run only if flag == 0 (decided on the
    device when they execute, glorie_ba_set_gate) and a call that ran increments hits."""
    L.need_cuda(poses, disps, intrinsics, targets, weights, ii, jj)
    L.need_contiguous(targets=targets, weights=weights, poses=poses, disps=disps,
                      intrinsics=intrinsics, disps_sens=disps_sens, ii=ii, jj=jj)
    _i64(ii, "ii"), _i64(jj, "jj")
    if eta is not None and not eta.is_contiguous():
        eta = eta.contiguous()  # the reference does not check eta (droid.cpp:107-114)
    B, h, w = disps.shape
    N = ii.shape[0]
    P = t1 - t0
    M = eta.shape[0] if eta is not None else 0
    if motion_only and eta is None:
        M = 0
    ctx = ctx or L.default_context()
    # [dx, dz] is the reference's return value; its one caller drops it (depth_video.py:217) -- callers
    # that do so here pass want_updates=False and save two fill launches per BA
    dx = torch.zeros((max(P, 0), 6), dtype=torch.float32, device=poses.device) if want_updates else None
    dz = torch.zeros((M, h * w), dtype=torch.float32, device=poses.device) if want_updates else None
    if disps_sens is not None and (disps_sens.shape != disps.shape):
        raise RuntimeError("disps_sens must have the shape of disps")
    if motion_only and M == 0:
        # the device still needs the slot count to carve its tables
        M = int(torch.unique(torch.cat([torch.arange(t0, t1, device=ii.device), ii])).numel())
    lib = L.load()
    if gate is not None:
        L.check(lib.glorie_ba_set_gate(ctx.handle, L.ptr(gate[0]), L.ptr(gate[1])), "glorie_ba_set_gate")
    try:
        L.check(lib.glorie_ba(ctx.handle, L.ptr(poses), L.ptr(disps), L.ptr(intrinsics),
                              L.ptr(disps_sens), L.ptr(targets), L.ptr(weights), L.ptr(eta),
                              L.ptr(ii), L.ptr(jj), B, N, M, h, w, int(t0), int(t1),
                              int(iterations), float(lm), float(ep),
                              int(bool(motion_only)) | (L.BA_TARGETS_HWC if targets_hwc else 0),
                              int(bool(depth_only)), L.ptr(dx),
                              L.ptr(dz) if (dz is not None and dz.numel()) else None,
                              L.stream_ptr()), "glorie_ba")
    finally:
        if gate is not None:
            lib.glorie_ba_set_gate(ctx.handle, None, None)
    if _CHECK_STATUS:
        # opt-in (GLORIE_CHECK_STATUS=1): surface what the reference raises on the host - a shape mismatch of
        # eta (droid_kernels.cu:1339-1352) - and report Cholesky failures; costs a device synchronisation
        st = ctx.ba_status()
        if st[0] & 1:
            raise RuntimeError(f"ba: eta has {M} rows but unique(cat(arange(t0, t1), ii)) has {st[1]} frames")
        if st[0] & 4:
            import warnings
            warnings.warn(f"ba: {st[2]} Gauss-Newton iteration(s) had a non positive definite system (zero update)")
    return [dx, dz] if want_updates else []
