"""ORACLE (test infrastructure only -- never imported by the product path).

Composition of the oracle pieces into ONE `FactorGraph.update()` call, stage by stage, in the
reference's own order of operations:

  FactorGraph.update            /root/reference/src/factor_graph.py:212-256
      reproject                 oracle.geom.reproject          (projective_ops.py:96-125)
      motion features           factor_graph.py:219-221        (clamp +-64)
      CorrBlock.__call__        oracle.corr.corr_lookup_pyramid (corr.py:43-53)
      update operator           supplied by the caller (a plain fp32 torch module whose weights and
                                outputs are pinned to the reference by tests/golden/update_module.npz)
      t0 rule                   factor_graph.py:229-230        t0 = max(1, ii.min() + 1)
      target / weight / damping factor_graph.py:232-248        damping = 0.2 * damping[unique(ii)] + EP
      use_inactive              factor_graph.py:237-243        inactive factors touching t0-3.. prepended
      DepthVideo.ba dispatcher  depth_video.py:287-296         stage 2 failure -> stage 1
      stage 1 "pose_depth"      oracle.ba.ba                   (droid_kernels.cu:1314-1437), clamp 1e-5
      stage 2 "depth_scale"     depth_video.py:222-285         valid mask, align_scale_and_shift,
                                                               mono_thres edge filter, itrs x oracle.dspo
      upsample                  oracle.geom.cvx_upsample       (depth_video.py:140-144)
      age += 1                  factor_graph.py:256

CorrBlock.__init__ under autocast (corr.py:26-41,67-76) is `corr_pyramid_fp16`.

Parity status: a composition of pieces that are pinned (update operator, pyramid, upsampling,
schur_solve, align_scale_and_shift) or "parity unpinned" against the native build (BA, lookup,
reproject) as stated in their own files; the bookkeeping between them is restated from the cited lines.
"""
import numpy as np

from . import ba as oba, corr as ocorr, dspo as odspo, geom as ogeom

F = np.float32


def corr_pyramid_fp16(fmap1, fmap2, num_levels=4):
    """CorrBlock.__init__ as the reference runs it (inside torch.autocast, corr.py:26-41,67-76):
    fp16 feature maps divided by 4 (rounded to fp16), fp16 GEMM with fp32 accumulation rounded to fp16,
    then avg_pool2d level by level on fp16 tensors (fp32 average rounded to fp16).
    fmap [N,C,h,w] -> levels [N,h,w,h>>l,w>>l] fp16"""
    N, C, h, w = fmap1.shape
    a = (np.asarray(fmap1, np.float16).astype(F) / F(4)).astype(np.float16).astype(F).reshape(N, C, h * w)
    b = (np.asarray(fmap2, np.float16).astype(F) / F(4)).astype(np.float16).astype(F).reshape(N, C, h * w)
    corr = np.empty((N, h * w, h * w), np.float16)
    for n in range(N):                                  # fp32 accumulation like the fp16 GEMM of the GPU
        corr[n] = (a[n].T @ b[n]).astype(np.float16)
    corr = corr.reshape(N, h, w, h, w)
    levels = []
    for _ in range(num_levels):
        levels.append(corr)
        corr = ocorr.avg_pool2(corr.astype(F)).astype(np.float16)
    return levels


def align_scale_and_shift(prediction, target, weights):
    """/root/reference/src/utils/common.py:401-437 in float32 (pinned by tests/golden/align.npz)"""
    p = np.asarray(prediction, F)
    t = np.asarray(target, F)
    wt = np.asarray(weights).astype(F)
    s = lambda x: x.reshape(x.shape[0], -1).astype(np.float64).sum(1).astype(F)
    a00, a01, a11 = s(wt * p * p), s(wt * p), s(wt)
    b0, b1 = s(wt * p * t), s(wt * t)
    with np.errstate(divide="ignore", invalid="ignore"):
        det = a00 * a11 - a01 * a01
        scale = (a11 * b0 - a01 * b1) / det
        shift = (-a01 * b0 + a00 * b1) / det
        err = np.abs(scale[:, None, None] * p + shift[:, None, None] - t)
        avg = s(err * wt) / s(wt)
    return scale.astype(F), shift.astype(F), avg.astype(F)


def valid_depth_mask(poses, disps, intrinsics0, index, mv_thresh, visible_num):
    """DepthVideo.update_valid_depth_mask (depth_video.py:326-361) on the map stack `disps`"""
    d = np.asarray(disps, F)[index]
    with np.errstate(divide="ignore"):
        depths = (F(1) / d).astype(F)
    thresh = (F(mv_thresh) * depths.reshape(len(index), -1).astype(np.float64).mean(1)).astype(F)
    count = ogeom.depth_filter(poses, disps, intrinsics0, index, thresh)
    depths = np.where(count >= visible_num, depths, F(np.nan))
    masks = np.zeros(depths.shape, bool)
    for b in range(len(index)):
        v = np.sort(depths[b][~np.isnan(depths[b])])
        med = v[(len(v) - 1) // 2] if len(v) else F(np.nan)      # torch.nanmedian: the lower median
        with np.errstate(invalid="ignore"):
            masks[b] = depths[b] < F(3) * med
    return masks


def depth_scale_stage(st, target, weight, eta, ii, jj, itrs, lm, ep, cfg):
    """DepthVideo.dspo(opt_type='depth_scale') (depth_video.py:222-285).  Mutates st; returns success."""
    n = st["n"]
    st["valid_small"][:n] = valid_depth_mask(st["poses"], st["disps"], st["intrinsics"][0], np.arange(n),
                                             cfg["mv_thresh"], cfg["visible_num"])
    mono_d, est_d, valid_d = st["mono_disps"][:n], st["disps"][:n], st["valid_small"][:n]
    scale_t, shift_t, error_t = align_scale_and_shift(mono_d, est_d, valid_d)
    avg = est_d.reshape(n, -1).astype(np.float64).mean(1).astype(F)
    st["depth_scale"][:n] = scale_t
    st["depth_shift"][:n] = shift_t
    ii = np.asarray(ii)
    jj = np.asarray(jj)
    keep = np.ones(len(ii), bool)
    eta_t = np.asarray(eta, F)
    if cfg["mono_thres"]:
        hw = valid_d.shape[1] * valid_d.shape[2]
        with np.errstate(invalid="ignore", divide="ignore"):
            bad = (error_t / avg > F(cfg["mono_thres"])) | np.isnan(error_t) | (scale_t < 0) | \
                  (valid_d.reshape(n, -1).sum(1) < hw * 0.5)
        for idx in np.nonzero(bad)[0]:
            keep &= ~((ii == idx) | (jj == idx))
        uq = np.unique(ii)
        uq_t = np.unique(ii[keep])
        eta_t = eta_t[np.isin(uq, uq_t)]
    st["edge_on"] = keep
    ii_t, jj_t = ii[keep], jj[keep]
    success = False
    for _ in range(itrs):
        if n > 0 and len(ii_t) > 0:
            d, s, q, _ = odspo.ba_with_scale_shift(target[keep], weight[keep], eta_t, st["poses"], st["disps"],
                                                   st["intrinsics"], ii_t, jj_t, st["mono_disps"],
                                                   st["depth_scale"], st["depth_shift"], st["valid_small"],
                                                   lm=lm, ep=ep, alpha=0.01)
            st["disps"], st["depth_scale"], st["depth_shift"] = d, s, q
            success = True
    st["disps"] = np.maximum(st["disps"], F(1e-5))
    return success


def pose_depth_stage(st, target, weight, eta, ii, jj, t0, t1, itrs, lm, ep, motion_only):
    """DepthVideo.dspo(opt_type='pose_depth') (depth_video.py:214-220): [N,h,w,2] -> [N,2,h,w], ba, clamp"""
    tg = np.ascontiguousarray(np.asarray(target, F).transpose(0, 3, 1, 2))
    wg = np.ascontiguousarray(np.asarray(weight, F).transpose(0, 3, 1, 2))
    p, d, dx, dz, info = oba.ba(st["poses"], st["disps"], st["intrinsics"][0], tg, wg, eta, ii, jj, t0, t1,
                                itrs, lm, ep, motion_only=motion_only)
    st["poses"], st["disps"] = p, np.maximum(d, F(1e-5))
    st["ba_info"] = info
    return True


def video_ba(st, target, weight, eta, ii, jj, t0, t1, itrs, lm, ep, motion_only, opt_type, cfg):
    """DepthVideo.ba (depth_video.py:287-296) for BA_type == 'DSPO'"""
    if opt_type == "depth_scale":
        ok = depth_scale_stage(st, target, weight, eta, ii, jj, itrs, lm, ep, cfg)
        if ok:
            return "depth_scale"
    pose_depth_stage(st, target, weight, eta, ii, jj, t0, t1, itrs, lm, ep, motion_only)
    return "pose_depth"


def update_step(st, gr, update_fn, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False,
                opt_type="pose_depth", cfg=None):
    """One FactorGraph.update() (factor_graph.py:212-256).

    st: video state -- poses [B,7], disps [B,h,w], disps_up [B,8h,8w], intrinsics [B,4], mono_disps,
        depth_scale, depth_shift [B], valid_small [B,h,w] bool, n (= counter.value)
    gr: graph state -- ii, jj [N], net [N,128,h,w], inp [N,128,h,w], target, weight [N,h,w,2],
        damping [B,h,w], age [N], pyramid (list of [N,h,w,h>>l,w>>l] fp16) and, for use_inactive,
        ii_inac, jj_inac, target_inac, weight_inac
    update_fn(net, inp, corr[N,196,h,w], motn[N,4,h,w], ii, jj) -> (net, delta[N,h,w,2], weight[N,h,w,2],
        eta[M,h,w], upmask[M,576,h,w])
    Returns a dict of the intermediate quantities (for stage-wise comparison)."""
    cfg = cfg or dict(mv_thresh=0.01, visible_num=2, mono_thres=0.1)
    ii, jj = np.asarray(gr["ii"], np.int64), np.asarray(gr["jj"], np.int64)
    _, h, w = st["disps"].shape
    coords1, _ = ogeom.reproject(st["poses"], st["disps"], st["intrinsics"], ii, jj)
    y, x = np.meshgrid(np.arange(h, dtype=F), np.arange(w, dtype=F), indexing="ij")
    coords0 = np.stack([x, y], -1)
    motn = np.concatenate([coords1 - coords0, gr["target"] - coords1], -1)
    motn = np.clip(motn.transpose(0, 3, 1, 2), F(-64), F(64)).astype(F)
    corr = ocorr.corr_lookup_pyramid(gr["pyramid"], np.ascontiguousarray(coords1.transpose(0, 3, 1, 2)), 3)
    net, delta, weight, eta, upmask = update_fn(gr["net"], gr["inp"], corr, motn, ii, jj)
    if t0 is None:
        t0 = max(1, int(ii.min()) + 1)
    gr["net"] = net
    gr["target"] = (coords1 + np.asarray(delta, F)).astype(F)
    gr["weight"] = np.asarray(weight, F)
    uniq = np.unique(ii)
    gr["damping"][uniq] = np.asarray(eta, F)
    if use_inactive:
        m = (gr["ii_inac"] >= t0 - 3) & (gr["jj_inac"] >= t0 - 3)
        ii_b = np.concatenate([gr["ii_inac"][m], ii])
        jj_b = np.concatenate([gr["jj_inac"][m], jj])
        target = np.concatenate([gr["target_inac"][m], gr["target"]], 0)
        wgt = np.concatenate([gr["weight_inac"][m], gr["weight"]], 0)
    else:
        ii_b, jj_b, target, wgt = ii, jj, gr["target"], gr["weight"]
    damping = (F(0.2) * gr["damping"][np.unique(ii_b)] + F(EP)).astype(F)
    if t1 is None:
        t1 = int(max(ii_b.max(), jj_b.max())) + 1
    stage = video_ba(st, target, wgt, damping, ii_b, jj_b, t0, t1, itrs, 1e-4, 0.1, motion_only, opt_type, cfg)
    # outside autocast the softmax of an fp16 mask is rounded to fp16 (droid_net.py:9-23 on a half tensor)
    half = np.asarray(upmask).dtype == np.float16
    st["disps_up"][uniq] = ogeom.cvx_upsample(st["disps"][uniq], upmask, np.float16 if half else None)
    gr["age"] = gr["age"] + 1
    return dict(coords1=coords1, motn=motn, corr=corr, delta=delta, eta=eta, upmask=upmask, damping=damping,
                t0=t0, t1=t1, stage=stage, ii=ii_b, jj=jj_b, target=target, weight=wgt)
