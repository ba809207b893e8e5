"""ORACLE (test infrastructure only -- never imported by the product path).

Brute-force exact k-NN, the pinned semantics of the build's neighbour search
(/root/reference/src/neural_point.py:264-313 calls faiss-gpu 1.7.2 IndexIVFFlat, an
approximate index whose k-means cannot be reproduced here -> "parity unpinned" against faiss;
see SURVEY.md section 8c).  Squared L2 distance evaluated as ((dx*dx + dy*dy) + dz*dz) in
float32, results ordered by (distance, index); missing results I = -1, D = FLT_MAX (faiss
convention).
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def knn_bruteforce(points, queries, k, chunk=2048):
    points = np.asarray(points, np.float32).reshape(-1, 3)
    queries = np.asarray(queries, np.float32).reshape(-1, 3)
    Q, n = queries.shape[0], points.shape[0]
    D = np.full((Q, k), FLT_MAX, np.float32)
    I = np.full((Q, k), -1, np.int64)
    if n == 0:
        return D, I
    kk = min(k, n)
    for s in range(0, Q, chunk):
        q = queries[s:s + chunk]
        d = q[:, None, :] - points[None]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        if n > 4 * kk:
            cand = np.argpartition(d2, kk - 1, axis=1)[:, :kk]
            # ties at the k-th distance: widen to everything <= k-th value
            kth = np.take_along_axis(d2, cand, 1).max(1)
            order = np.empty((q.shape[0], kk), np.int64)
            for r in range(q.shape[0]):
                c = np.nonzero(d2[r] <= kth[r])[0]
                c = c[np.lexsort((c, d2[r, c]))][:kk]
                order[r] = c
        else:
            order = np.lexsort((np.broadcast_to(np.arange(n), d2.shape), d2), axis=1)[:, :kk]
        D[s:s + chunk, :kk] = np.take_along_axis(d2, order, 1)
        I[s:s + chunk, :kk] = order
    return D, I


def build_kdtree(points):
    """k-d tree over the cloud for knn_kdtree (scipy; the CPU baseline of bench.py - the parity checker stays
    knn_bruteforce, whose (distance, index) tie order the k-d tree does not promise)"""
    from scipy.spatial import cKDTree
    return cKDTree(np.asarray(points, np.float64).reshape(-1, 3))


def knn_kdtree(tree, queries, k, workers=-1):
    """exact k nearest neighbours through a k-d tree, all cores (workers=-1) -> (squared distances float32, indices int64)
    in the brute-force search's conventions (missing: FLT_MAX / -1)"""
    q = np.asarray(queries, np.float64).reshape(-1, 3)
    d, i = tree.query(q, k=k, workers=workers)
    d, i = d.reshape(q.shape[0], k), i.reshape(q.shape[0], k).astype(np.int64)
    miss = ~np.isfinite(d)
    D = np.where(miss, FLT_MAX, (d * d)).astype(np.float32)
    I = np.where(miss, -1, i)
    return D, I


def neighbor_count(D, radius):
    r = np.asarray(radius, np.float32)
    r2 = (r * r).reshape(-1, 1) if r.ndim else r * r
    return (D < r2).sum(-1).astype(np.int32)


def idw_gather(D, I, nn, feats, radius, min_nn=2):
    """decoder.py:148-171 -> c [Q,C], has [Q]"""
    D = np.asarray(D, np.float32)
    r = np.asarray(radius, np.float32)
    r2 = (r * r).reshape(-1, 1) if r.ndim else r * r
    w = np.float32(1.0) / (D + np.float32(1e-10))
    w = np.where((D > r2) | (I < 0), np.float32(0), w).astype(np.float32)
    w = w / np.maximum(np.abs(w).sum(1, keepdims=True), np.float32(1e-12))
    c = (w[..., None] * feats[np.clip(I, 0, None)]).sum(1).astype(np.float32)
    has = nn > min_nn - 1
    c[~has] = 0
    return c, has, w


def composite(raw, z, coef=0.1):
    """common.py:261-299 in float64"""
    raw = np.asarray(raw, np.float64)
    z = np.asarray(z, np.float64)
    alpha = 1.0 / (1.0 + np.exp(-coef * raw[..., 3]))
    T = np.cumprod(np.concatenate([np.ones((alpha.shape[0], 1)), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    ws = w.sum(-1, keepdims=True) + 1e-10
    rgb = (w[..., None] * raw[..., :3]).sum(-2) / ws
    depth = (w * z).sum(-1) / ws[:, 0]
    var = (w * (z - depth[:, None]) ** 2).sum(-1)
    return depth, var, rgb, w
