"""Pins the oracle's SE3 helpers with closed-form identities (no GPU)."""
import numpy as np
from scipy.spatial.transform import Rotation

from oracle import se3


def _rand_pose(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.standard_normal(3), q]).astype(np.float32)


def test_exp_matches_scipy():
    rng = np.random.default_rng(0)
    for scale in (1e-6, 1e-3, 0.3, 2.0):
        phi = (rng.standard_normal(3) * scale).astype(np.float32)
        q = se3.so3_exp(phi)
        np.testing.assert_allclose(q, Rotation.from_rotvec(phi.astype(np.float64)).as_quat(), atol=2e-6)


def test_se3_exp_translation_is_V_tau():
    rng = np.random.default_rng(1)
    xi = (rng.standard_normal(6) * 0.4).astype(np.float32)
    t, q = se3.se3_exp(xi)
    phi = xi[3:].astype(np.float64)
    th = np.linalg.norm(phi)
    K = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
    V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    np.testing.assert_allclose(t, V @ xi[:3], atol=1e-6)


def test_rel_pose_and_act_match_matrices():
    rng = np.random.default_rng(2)
    pi, pj = _rand_pose(rng), _rand_pose(rng)
    t, q = se3.rel_pose(pi, pj)
    Tij = se3.matrix(pj) @ np.linalg.inv(se3.matrix(pi))
    np.testing.assert_allclose(se3.matrix(np.concatenate([t, q])), Tij, atol=2e-5)
    X = rng.standard_normal((5, 4)).astype(np.float32)
    np.testing.assert_allclose(se3.act(t, q, X), X @ Tij.T, atol=2e-5)


def test_retract_is_left_multiplication():
    rng = np.random.default_rng(3)
    p = _rand_pose(rng)
    xi = (rng.standard_normal(6) * 0.1).astype(np.float32)
    dt, dq = se3.se3_exp(xi)
    got = se3.matrix(se3.retract(xi, p))
    np.testing.assert_allclose(got, se3.matrix(np.concatenate([dt, dq])) @ se3.matrix(p), atol=2e-5)


def test_adjT_is_dual_adjoint():
    """Ji = -adjT(Gij, Jj): perturbing pose i by xi moves Gij = Gj Gi^-1 to Gij exp(-xi),
    which equals exp(-Ad_Gij xi) Gij, so a row J (wrt left perturbation of Gij) maps to
    -J Ad_Gij.  adjT(J) must therefore equal J @ Ad(Gij)."""
    rng = np.random.default_rng(4)
    g = _rand_pose(rng)
    T = se3.matrix(g)
    R, t = T[:3, :3], T[:3, 3]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ad = np.block([[R, tx @ R], [np.zeros((3, 3)), R]])
    J = rng.standard_normal(6).astype(np.float32)
    np.testing.assert_allclose(se3.adjT(g[:3], g[3:], J), J @ Ad, atol=2e-5)
