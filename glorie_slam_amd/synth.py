"""Seeded synthetic inputs of the benchmark.

Everything is generated with numpy on the host (seed 43 = `setup_seed` of
configs/mono_point_slam.yaml:7) so that the GPU path, the oracle and the bench all see
bit-identical inputs.  No dataset or checkpoint is needed.
"""
import math

import numpy as np

SEED = 43


def camera(h=60, w=80):
    """BA-resolution intrinsics [fx fy cx cy]: the 640x480 camera (fx=fy=320, cx=319.5,
    cy=239.5) resized to (8h x 8w) and divided by 8 (datasets.py:85-96 scaling rule)."""
    sx, sy = (8.0 * w) / 640.0, (8.0 * h) / 480.0
    return np.array([320.0 * sx / 8.0, 320.0 * sy / 8.0, 319.5 * sx / 8.0, 239.5 * sy / 8.0],
                    np.float32)


def _quat_y(theta):
    return np.array([0.0, math.sin(theta / 2), 0.0, math.cos(theta / 2)], np.float32)


def keyframe_graph(K=8, h=60, w=80, radius=3, seed=SEED, noise_px=0.5, buffer=None):
    """Graph G8 of BASELINE.md: K keyframes on a gentle arc, bootstrap topology |i-j|<=radius.

    Returns a dict of numpy arrays: poses [B,7], disps [B,h,w], intrinsics [B,4], ii, jj [N]
    int64, eta [K,h,w], weight [N,2,h,w], and `noise` [N,2,h,w] to be added to the
    reprojection to form the BA targets."""
    rng = np.random.default_rng(seed)
    B = buffer or K
    poses = np.zeros((B, 7), np.float32)
    poses[:, 6] = 1.0
    for k in range(K):
        poses[k, :3] = (0.05 * k, 0.01 * k, 0.0)
        poses[k, 3:] = _quat_y(0.02 * k)
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    disps = np.ones((B, h, w), np.float32)
    for k in range(K):
        depth = 2.0 + 0.5 * np.sin(2 * np.pi * (x + 3 * k) / w) * np.cos(2 * np.pi * (y + 2 * k) / h)
        disps[k] = (1.0 / depth + rng.uniform(-0.01, 0.01, (h, w))).astype(np.float32)
    intr = np.tile(camera(h, w)[None], (B, 1)).astype(np.float32)
    ii, jj = [], []
    for i in range(K):
        for j in range(K):
            if i != j and abs(i - j) <= radius:
                ii.append(i)
                jj.append(j)
    ii = np.array(ii, np.int64)
    jj = np.array(jj, np.int64)
    N = len(ii)
    weight = rng.uniform(0.0, 1.0, (N, 2, h, w)).astype(np.float32)
    noise = rng.normal(0.0, noise_px, (N, 2, h, w)).astype(np.float32)
    eta = (0.2 * rng.uniform(1e-3, 2e-2, (K, h, w)) + 1e-7).astype(np.float32)
    return dict(poses=poses, disps=disps, intrinsics=intr, ii=ii, jj=jj, weight=weight,
                noise=noise, eta=eta, K=K, h=h, w=w)


def loop_graph(K=64, h=30, w=40, seed=SEED, per_frame=6):
    """Graph GL: K keyframes on a circle looking inwards; edges to temporal neighbours +-1..3
    plus loop edges to the frames half a turn of neighbours away (<= 6K edges, backend.py:57)."""
    rng = np.random.default_rng(seed)
    poses = np.zeros((K, 7), np.float32)
    for k in range(K):
        a = 2 * np.pi * k / K
        # world->camera: small circle of radius 0.3 m, yaw follows the circle slowly
        yaw = 0.15 * np.sin(a)
        poses[k, :3] = (0.3 * np.cos(a) - 0.3, 0.02 * np.sin(2 * a), 0.3 * np.sin(a))
        poses[k, 3:] = _quat_y(yaw)
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    disps = np.zeros((K, h, w), np.float32)
    for k in range(K):
        depth = 2.5 + 0.4 * np.sin(2 * np.pi * x / w + k * 0.1) * np.cos(2 * np.pi * y / h)
        disps[k] = (1.0 / depth + rng.uniform(-0.005, 0.005, (h, w))).astype(np.float32)
    intr = np.tile(camera(h, w)[None], (K, 1)).astype(np.float32)
    es = set()
    for i in range(K):
        for d in (1, 2, 3):
            if i - d >= 0:
                es.add((i, i - d))
                es.add((i - d, i))
        if i >= K // 2 + 21:  # loop-closure style edges (si - sj > 20)
            j = i - K // 2
            es.add((i, j))
            es.add((j, i))
    es = sorted(es)
    ii = np.array([e[0] for e in es], np.int64)
    jj = np.array([e[1] for e in es], np.int64)
    N = len(ii)
    weight = rng.uniform(0.0, 1.0, (N, 2, h, w)).astype(np.float32)
    noise = rng.normal(0.0, 0.5, (N, 2, h, w)).astype(np.float32)
    eta = (0.2 * rng.uniform(1e-3, 2e-2, (K, h, w)) + 1e-7).astype(np.float32)
    return dict(poses=poses, disps=disps, intrinsics=intr, ii=ii, jj=jj, weight=weight,
                noise=noise, eta=eta, K=K, h=h, w=w)


def feature_maps(K, h, w, seed=SEED, C=128):
    """fmaps ~ N(0,1) fp16 [K,1,C,h,w], nets = tanh(N(0,1)), inps = relu(N(0,1)) fp16"""
    rng = np.random.default_rng(seed + 1)
    fmaps = rng.standard_normal((K, 1, C, h, w)).astype(np.float16)
    nets = np.tanh(rng.standard_normal((K, C, h, w))).astype(np.float16)
    inps = np.maximum(rng.standard_normal((K, C, h, w)), 0).astype(np.float16)
    return fmaps, nets, inps


def box_cloud(n_hits=174762, seed=SEED, n_add=3, box=(6.0, 4.0, 3.0)):
    """Cloud PC: n_hits surface points on the inside of a box centred at the origin, each
    replicated at 0.95/1.0/1.05 of its distance from the centre (N_add = 3) + N(0, 5 mm)."""
    rng = np.random.default_rng(seed + 2)
    bx, by, bz = box
    areas = np.array([by * bz, by * bz, bx * bz, bx * bz, bx * by, bx * by])
    face = rng.choice(6, size=n_hits, p=areas / areas.sum())
    u = rng.uniform(-0.5, 0.5, n_hits)
    v = rng.uniform(-0.5, 0.5, n_hits)
    pts = np.zeros((n_hits, 3), np.float64)
    for f in range(6):
        m = face == f
        ax = f // 2
        sgn = 1.0 if f % 2 == 0 else -1.0
        o = [a for a in range(3) if a != ax]
        pts[m, ax] = sgn * box[ax] / 2
        pts[m, o[0]] = u[m] * box[o[0]]
        pts[m, o[1]] = v[m] * box[o[1]]
    scales = np.linspace(0.95, 1.05, n_add)
    cloud = np.concatenate([pts * s for s in scales], 0)
    cloud = cloud + rng.normal(0.0, 0.005, cloud.shape)
    geo = rng.normal(0.0, 0.1, (cloud.shape[0], 32)).astype(np.float32)
    col = rng.normal(0.0, 0.1, (cloud.shape[0], 32)).astype(np.float32)
    return cloud.astype(np.float32), geo, col


def box_rays(H=480, W=640, box=(6.0, 4.0, 3.0), seed=SEED, fx=320.0, fy=320.0, cx=319.5, cy=239.5):
    """One camera at the box centre looking along +x (OpenGL convention of get_rays,
    common.py:302-322); gt_depth = exact ray-box z-depth; dynamic radius U(0.04,0.16)*depth/3."""
    rng = np.random.default_rng(seed + 3)
    j, i = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    dirs = np.stack([(i - cx) / fx, -(j - cy) / fy, -np.ones_like(i)], -1)  # camera frame
    # c2w: camera -z (view direction) -> world +x ; camera x -> world -y... choose a proper rotation
    R = np.array([[0.0, 0.0, -1.0],
                  [-1.0, 0.0, 0.0],
                  [0.0, 1.0, 0.0]], np.float32)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = R
    rays_d = (dirs[..., None, :] * R).sum(-1).reshape(-1, 3).astype(np.float32)
    rays_o = np.zeros_like(rays_d)
    half = np.array(box, np.float32) / 2
    with np.errstate(divide="ignore"):
        tmax = np.where(rays_d > 0, half / rays_d, np.where(rays_d < 0, -half / rays_d, np.inf))
    depth = tmax.min(-1).astype(np.float32)  # z-depth because |dirs_z| = 1
    radius = (rng.uniform(0.04, 0.16, depth.shape) * depth / 3.0).astype(np.float32)
    return rays_o, rays_d, depth, radius, c2w
