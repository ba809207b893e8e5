"""Pins the BA oracle (oracle/ba.py) without the native reference (which cannot be built
here): finite-difference Jacobians, Gauss-Newton fixed point / descent, and agreement of the
Schur-eliminated step with a dense solve of the full normal equations."""
import numpy as np

from oracle import ba as oba, geom as ogeom, se3
import glorie_slam_amd.synth as synth


def small_graph(K=4, h=12, w=16, noise=0.5, seed=5):
    g = synth.keyframe_graph(K=K, h=h, w=w, radius=2, seed=seed, noise_px=noise)
    coords, _ = ogeom.reproject(g["poses"], g["disps"], g["intrinsics"], g["ii"], g["jj"])
    g["target"] = (coords.transpose(0, 3, 1, 2) + g["noise"]).astype(np.float32)
    return g


def test_jacobians_finite_difference():
    g = small_graph()
    intr = g["intrinsics"][0]
    i, j = 1, 2
    n = [k for k in range(len(g["ii"])) if g["ii"][k] == i and g["jj"][k] == j][0]
    T = oba.per_edge_terms(g["poses"], g["disps"], intr, g["target"][n], g["weight"][n], i, j)
    HW = g["h"] * g["w"]

    def resid(poses, disps):
        t = oba.per_edge_terms(poses, disps, intr, g["target"][n], g["weight"][n], i, j)
        return t["r"].astype(np.float64)  # [2,HW]

    r0 = resid(g["poses"], g["disps"])
    eps = 1e-3
    # v_j = sum w r J_j  => compare with -d(0.5 sum w r^2)/d xi_j
    w = T["w"].astype(np.float64)
    for which, col in ((j, 1), (i, 0)):
        grad = np.zeros(6)
        for a in range(6):
            xi = np.zeros(6, np.float32)
            xi[a] = eps
            pp = g["poses"].copy()
            pp[which] = se3.retract(xi, pp[which])
            rp = resid(pp, g["disps"])
            xi[a] = -eps
            pm = g["poses"].copy()
            pm[which] = se3.retract(xi, pm[which])
            rm = resid(pm, g["disps"])
            grad[a] = -0.5 * ((w * rp ** 2).sum() - (w * rm ** 2).sum()) / (2 * eps)
        scale = np.abs(grad).max()
        np.testing.assert_allclose(T["vs"][col], grad, atol=2e-2 * scale)
    # disparity Jacobian: bz = sum_c w r Jz
    d = g["disps"].copy()
    d[i] += eps
    rp = resid(g["poses"], d)
    d[i] -= 2 * eps
    rm = resid(g["poses"], d)
    gz = -0.5 * ((w * rp ** 2).sum(0) - (w * rm ** 2).sum(0)) / (2 * eps)
    np.testing.assert_allclose(T["bz"], gz, atol=2e-2 * np.abs(gz).max())


def test_noise_free_targets_are_a_fixed_point():
    g = small_graph(noise=0.0)
    p, d, dx, dz, info = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                                g["eta"], g["ii"], g["jj"], 1, g["K"], 2, 1e-4, 0.1)
    assert info["failed"] == 0
    assert np.abs(dx).max() < 1e-4 and np.abs(dz).max() < 1e-4


def test_gauss_newton_step_reduces_cost():
    g = small_graph(noise=0.0)
    rng = np.random.default_rng(0)
    poses = g["poses"].copy()
    for k in range(1, g["K"]):
        poses[k] = se3.retract((rng.standard_normal(6) * 0.01).astype(np.float32), poses[k])
    disps = (g["disps"] * (1 + 0.03 * rng.standard_normal(g["disps"].shape))).astype(np.float32)
    intr = g["intrinsics"][0]
    c0 = oba.reprojection_cost(poses, disps, intr, g["target"], g["weight"], g["ii"], g["jj"])
    p1, d1, *_ = oba.ba(poses, disps, intr, g["target"], g["weight"], g["eta"], g["ii"], g["jj"],
                        1, g["K"], 2, 1e-4, 0.1)
    c1 = oba.reprojection_cost(p1, d1, intr, g["target"], g["weight"], g["ii"], g["jj"])
    assert c1 < 0.2 * c0


def test_schur_step_equals_full_normal_equations():
    """dx from the reduced system == the pose part of the dense (pose + depth) solve."""
    g = small_graph()
    intr = g["intrinsics"][0]
    K, h, w = g["K"], g["h"], g["w"]
    HW = h * w
    t0, t1 = 1, K
    P = t1 - t0
    lm, ep = 1e-4, 0.1
    ii, jj = [int(v) for v in g["ii"]], [int(v) for v in g["jj"]]
    kx = sorted(set(list(range(t0, t1)) + ii))
    M = len(kx)
    nz = M * HW
    Hpp = np.zeros((6 * P, 6 * P)); Hpz = np.zeros((6 * P, nz)); Hzz = np.zeros(nz)
    bp = np.zeros(6 * P); bz = np.zeros(nz)
    for n in range(len(ii)):
        T = oba.per_edge_terms(g["poses"], g["disps"], intr, g["target"][n], g["weight"][n], ii[n], jj[n])
        i, j = ii[n] - t0, jj[n] - t0
        k = kx.index(ii[n])
        blocks = {(0, 0): T["Hs"][0], (0, 1): T["Hs"][1], (1, 0): T["Hs"][2], (1, 1): T["Hs"][3]}
        for (a, pa) in ((0, i), (1, j)):
            if pa < 0:
                continue
            bp[6 * pa:6 * pa + 6] += T["vs"][a]
            E = T["Eii"] if a == 0 else T["Eij"]
            Hpz[6 * pa:6 * pa + 6, k * HW:(k + 1) * HW] += E
            for (b, pb) in ((0, i), (1, j)):
                if pb >= 0:
                    Hpp[6 * pa:6 * pa + 6, 6 * pb:6 * pb + 6] += blocks[(a, b)]
        Hzz[k * HW:(k + 1) * HW] += T["Cii"]
        bz[k * HW:(k + 1) * HW] += T["bz"]
    Hzz += g["eta"][:M].reshape(-1)
    # damping after Schur == add (ep + lm*diag(S)) to the pose diagonal of the reduced system
    S = Hpp - (Hpz / Hzz) @ Hpz.T
    dgl = np.diag(S).copy()
    S[np.diag_indices_from(S)] = dgl + ep + lm * dgl
    dx_full = np.linalg.solve(S, bp - (Hpz / Hzz) @ bz).reshape(P, 6)
    _, _, dx, dz, info = oba.ba(g["poses"], g["disps"], intr, g["target"], g["weight"], g["eta"][:M],
                                g["ii"], g["jj"], t0, t1, 1, lm, ep)
    np.testing.assert_allclose(dx, dx_full, rtol=2e-3, atol=2e-6)
