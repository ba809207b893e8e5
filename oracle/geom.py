"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatements (numpy float32) of the projective-geometry ops:
  * reproject        /root/reference/src/geom/projective_ops.py:18-125 (python path, MIN_DEPTH 0.2)
  * frame_distance   /root/reference/src/lib/droid_kernels.cu:518-657
  * iproj            /root/reference/src/lib/droid_kernels.cu:779-850
  * depth_filter     /root/reference/src/lib/droid_kernels.cu:661-775
  * cvx_upsample     /root/reference/src/modules/droid_net/droid_net.py:9-23

Parity status: native kernels "parity unpinned" (cannot be compiled here); cvx_upsample is
pinned by the golden fixture generated from the importable reference function
(tests/golden/cvx_upsample.npz).
"""
import numpy as np

from . import se3

F = np.float32
STEREO_T = np.array([-0.1, 0.0, 0.0], F)
STEREO_Q = np.array([0.0, 0.0, 0.0, 1.0], F)


def _grid(h, w):
    y, x = np.meshgrid(np.arange(h, dtype=F), np.arange(w, dtype=F), indexing="ij")
    return x, y


def _rel(poses, i, j):
    if i == j:
        return STEREO_T, STEREO_Q
    return se3.rel_pose(poses[i], poses[j])


def reproject(poses, disps, intrinsics, ii, jj):
    """-> coords [N,h,w,2], valid [N,h,w,1] (projective_ops.py:96-125 with jacobian=False)"""
    poses = np.asarray(poses, F)
    disps = np.asarray(disps, F)
    intrinsics = np.asarray(intrinsics, F)
    N = len(ii)
    _, h, w = disps.shape
    x, y = _grid(h, w)
    coords = np.zeros((N, h, w, 2), F)
    valid = np.zeros((N, h, w, 1), F)
    for n, (i, j) in enumerate(zip(ii, jj)):
        i, j = int(i), int(j)
        fxi, fyi, cxi, cyi = intrinsics[i]
        fxj, fyj, cxj, cyj = intrinsics[j]
        X0 = np.stack([(x - cxi) / fxi, (y - cyi) / fyi, np.ones_like(x), disps[i]], -1).astype(F)
        t, q = _rel(poses, i, j)
        X1 = se3.act(t, q, X0)
        Z = np.where(X1[..., 2] < F(0.1), F(1), X1[..., 2])
        d = F(1) / Z
        coords[n, ..., 0] = fxj * (X1[..., 0] * d) + cxj
        coords[n, ..., 1] = fyj * (X1[..., 1] * d) + cyj
        valid[n, ..., 0] = ((X1[..., 2] > F(0.2)) & (X0[..., 2] > F(0.2))).astype(F)
    return coords, valid


def _tree_sum_256(v):
    """blockReduce of droid_kernels.cu:45-55 on 256 per-thread partial sums (fp32 order)"""
    s = np.asarray(v, F).copy()
    s[:128] += s[128:256]
    s[:64] += s[64:128]
    for off in (32, 16, 8, 4, 2, 1):
        # warpReduce: threads < 32 do s[t] += s[t+off] sequentially with volatile semantics
        s[:32] = s[:32] + s[off:off + 32]
    return s[0]


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """-> dist [K].  Thread t of the 256-thread block owns pixels t, t+256, ...; per-thread
    accumulation order and the tree reduction follow the kernel so the fp32 result (which
    the graph topology is thresholded on) is reproduced as closely as arithmetic allows."""
    poses = np.asarray(poses, F)
    disps = np.asarray(disps, F)
    fx, fy, cx, cy = np.asarray(intrinsics, F)[:4]
    _, h, w = disps.shape
    HW = h * w
    x, y = _grid(h, w)
    u = x.reshape(-1)
    v = y.reshape(-1)
    beta = F(beta)
    out = np.zeros(len(ii), F)
    npad = (HW + 255) // 256 * 256
    for n, (i, j) in enumerate(zip(ii, jj)):
        i, j = int(i), int(j)
        t, q = se3.rel_pose(poses[i], poses[j])
        Xi = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u), disps[i].reshape(-1)], -1).astype(F)
        Xj = se3.act(t, q, Xi)
        with np.errstate(divide="ignore", invalid="ignore"):
            du = fx * (Xj[:, 0] / Xj[:, 2]) + cx - u
            dv = fy * (Xj[:, 1] / Xj[:, 2]) + cy - v
            d1 = np.sqrt(du * du + dv * dv).astype(F)
            ok1 = Xj[:, 2] > F(0.25)
            X2 = Xi[:, :3] + Xi[:, 3:4] * t
            du = fx * (X2[:, 0] / X2[:, 2]) + cx - u
            dv = fy * (X2[:, 1] / X2[:, 2]) + cy - v
            d2 = np.sqrt(du * du + dv * dv).astype(F)
            ok2 = X2[:, 2] > F(0.25)

        def per_thread(a1, a2):
            """interleaved accumulation a1[k], a2[k] for k = t, t+256, ... per thread"""
            a = np.zeros((npad, 2), F)
            a[:HW, 0] = a1
            a[:HW, 1] = a2
            a = a.reshape(-1, 256, 2)
            acc = np.zeros(256, F)
            for r in range(a.shape[0]):
                acc = acc + a[r, :, 0]
                acc = acc + a[r, :, 1]
            return acc

        accum = per_thread(np.where(ok1, beta * d1, F(0)), np.where(ok2, (F(1) - beta) * d2, F(0)))
        total = per_thread(np.full(HW, beta, F), np.full(HW, F(1) - beta, F))
        vld = per_thread(np.where(ok1, beta, F(0)), np.where(ok2, F(1) - beta, F(0)))
        A, T, V = _tree_sum_256(accum), _tree_sum_256(total), _tree_sum_256(vld)
        out[n] = F(1000.0) if V / (T + F(1e-8)) < F(0.75) else A / V
    return out


def iproj(poses, disps, intrinsics):
    poses = np.asarray(poses, F)
    disps = np.asarray(disps, F)
    fx, fy, cx, cy = np.asarray(intrinsics, F)[:4]
    num, h, w = disps.shape
    x, y = _grid(h, w)
    pts = np.zeros((num, h, w, 3), F)
    for b in range(num):
        Xi = np.stack([(x - cx) / fx, (y - cy) / fy, np.ones_like(x), disps[b]], -1).astype(F)
        Xj = se3.act(poses[b, :3], poses[b, 3:], Xi)
        with np.errstate(divide="ignore", invalid="ignore"):
            pts[b] = Xj[..., :3] / Xj[..., 3:4]
    return pts


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """count [num,h,w]: number of the 6 temporal neighbours (ix-1..-3, ix+3..+5) whose
    disparity agrees within thresh (droid_kernels.cu:693-773)"""
    poses = np.asarray(poses, F)
    disps = np.asarray(disps, F)
    fx, fy, cx, cy = np.asarray(intrinsics, F)[:4]
    B, h, w = disps.shape
    x, y = _grid(h, w)
    count = np.zeros((len(ix), h, w), F)
    for b, i in enumerate(ix):
        i = int(i)
        t_thr = np.float64(thresh[b])
        Xi = np.stack([(x - cx) / fx, (y - cy) / fy, np.ones_like(x), disps[i]], -1).astype(F)
        for nb in range(6):
            j = i - nb - 1 if nb < 3 else i + nb
            if j < 0 or j >= B:
                continue
            t, q = se3.rel_pose(poses[i], poses[j])
            Xj = se3.act(t, q, Xi)
            with np.errstate(divide="ignore", invalid="ignore"):
                uj = fx * (Xj[..., 0] / Xj[..., 2]) + cx
                vj = fy * (Xj[..., 1] / Xj[..., 2]) + cy
                dj = Xj[..., 3] / Xj[..., 2]
                u0 = np.floor(uj)
                v0 = np.floor(vj)
                ok = np.isfinite(u0) & np.isfinite(v0)
                u0i = np.where(ok, u0, -1).astype(np.int64)
                v0i = np.where(ok, v0, -1).astype(np.int64)
                inb = (u0i >= 0) & (v0i >= 0) & (u0i < w - 1) & (v0i < h - 1)
                uc = np.clip(u0i, 0, w - 2)
                vc = np.clip(v0i, 0, h - 2)
                zj = 1.0 / dj.astype(np.float64)
                hit = np.zeros((h, w), bool)
                for (oy, ox) in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    dd = disps[j][vc + oy, uc + ox].astype(np.float64)
                    hit |= np.abs(zj - 1.0 / dd) < t_thr
            count[b] += (inb & hit).astype(F)
    return count


def cvx_upsample(data, mask, round_softmax_to=None):
    """data [M,h,w] f32, mask [M,576,h,w] -> [M,8h,8w] (droid_net.py:9-23).
    round_softmax_to=np.float16 reproduces torch.softmax on a half tensor outside autocast
    (weights rounded to fp16 before the fp32 product)."""
    data = np.asarray(data, F)
    M, h, w = data.shape
    m = np.asarray(mask).astype(F).reshape(M, 9, 8, 8, h, w)
    m = m - m.max(1, keepdims=True)
    e = np.exp(m)
    sm = (e / e.sum(1, keepdims=True)).astype(F)
    if round_softmax_to is not None:
        sm = sm.astype(round_softmax_to).astype(F)
    pad = np.zeros((M, h + 2, w + 2), F)
    pad[:, 1:-1, 1:-1] = data
    taps = np.stack([pad[:, ky:ky + h, kx:kx + w] for ky in range(3) for kx in range(3)], 1)  # [M,9,h,w]
    up = (sm * taps[:, :, None, None]).sum(1)  # [M,8,8,h,w]
    return up.transpose(0, 3, 1, 4, 2).reshape(M, 8 * h, 8 * w).astype(F)
