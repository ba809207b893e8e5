"""ORACLE (test infrastructure only -- never imported by the product path).

Literal restatement (python loops over plain lists) of the graph-topology construction of
/root/reference/src/factor_graph.py:
    add_neighborhood_factors        :312-320
    add_proximity_factors           :323-383
    add_backend_proximity_factors   :386-462
given the frame-distance vector `d` (what video.distance returns).  Returns the edge list `es`
in insertion order (before the duplicate filter of add_factors).  torch.argsort ties are
broken by index (stable), which is the documented tie rule of the build.
"""
import numpy as np


def neighborhood(t0, t1, r=3):
    return [(i, j) for i in range(t0, t1) for j in range(t0, t1) if 0 < abs(i - j) <= r]


def proximity(d, t, existing, t0=0, t1=0, rad=2, nms=2, thresh=16.0, max_factors=-1):
    """d: distances for the meshgrid(arange(t0,t), arange(t1,t)) flattened row-major."""
    d = np.array(d, np.float32)
    ii = [i for i in range(t0, t) for _ in range(t1, t)]
    jj = [j for _ in range(t0, t) for j in range(t1, t)]
    for k in range(len(d)):
        if ii[k] - rad < jj[k]:
            d[k] = np.inf
        if d[k] > 100:
            d[k] = np.inf

    def suppress(i, j):
        for di in range(-nms, nms + 1):
            for dj in range(-nms, nms + 1):
                if abs(di) + abs(dj) <= max(min(abs(i - j) - 2, nms), 0):
                    i1, j1 = i + di, j + dj
                    if (t0 <= i1 < t) and (t1 <= j1 < t):
                        d[(i1 - t0) * (t - t1) + (j1 - t1)] = np.inf

    for (i, j) in existing:
        suppress(i, j)
    es = []
    for i in range(t0, t):
        for j in range(max(i - rad - 1, 0), i):
            es.append((i, j))
            es.append((j, i))
            d[(i - t0) * (t - t1) + (j - t1)] = np.inf
    for k in np.argsort(d, kind="stable"):
        if d[k] > thresh:
            continue
        if len(es) > max_factors:
            break
        i, j = ii[k], jj[k]
        es.append((i, j))
        es.append((j, i))
        suppress(i, j)
    return es


def backend_proximity(d, t_start, t_end, nms, radius, thresh, max_factors, t_start_loop=None, loop=False):
    if t_start_loop is None or not loop:
        t_start_loop = t_start
    ilen, jlen = t_end - t_start_loop, t_end - t_start
    d = np.array(d, np.float32).reshape(ilen, jlen)
    rawd = d.copy()
    for a in range(ilen):
        for b in range(jlen):
            i, j = a + t_start_loop, b + t_start
            if i - radius < j or d[a, b] > thresh:
                d[a, b] = np.inf
    es = []
    for i in range(t_start_loop, t_end):
        for j in range(max(i - radius - 1, 0), i):
            es.append((i, j))
            es.append((j, i))
            d[i - t_start_loop, j - t_start] = np.inf
    flat = d.reshape(-1)
    order = [k for k in np.argsort(flat, kind="stable") if flat[k] <= thresh]
    loop_edges = 0
    for k in order:
        di, dj = k // jlen, k % jlen
        if d[di, dj] > thresh:
            continue
        if len(es) > max_factors:
            break
        i, j = di + t_start_loop, dj + t_start
        if loop:
            sub = []
            for si in range(max(i - 1, t_start_loop), min(i + 2, t_end)):
                for sj in range(max(j - 1, t_start), min(j + 2, t_end)):
                    if rawd[si - t_start_loop, sj - t_start] <= thresh and si != sj and si - sj > 20:
                        sub.append((si, sj))
            es += sub
            loop_edges += len(sub)
        else:
            es += [(i, j), (j, i)]
        d[max(0, di - nms):min(ilen, di + nms + 1), max(0, dj - nms):min(jlen, dj + nms + 1)] = np.inf
    if len(es) < 3 or (loop and loop_edges == 0):
        return []
    return es
