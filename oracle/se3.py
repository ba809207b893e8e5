"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in numpy float32, of the SE3 device helpers that fix the conventions of
the native path: /root/reference/src/lib/droid_kernels.cu:58-175 (actSO3, actSE3, adjSE3,
relSE3, expSO3, expSE3) and :877-895 (retrSE3).

pose = [tx ty tz qx qy qz qw], tangent = [tau, phi], left retraction exp(xi) * G.

Parity status: the reference's CUDA cannot be compiled in this environment (no nvcc, Eigen
submodule empty) and the reference has no tests -> "parity unpinned" for these helpers; they
are pinned instead by closed-form identities in tests/test_oracle_se3.py (exp/log round trip
against scipy Rotation, group axioms, adjoint identity, finite differences).
"""
import numpy as np

F = np.float32


def quat_rotate(q, v):
    """droid_kernels.cu:58-68 -- q [...,4] (xyzw), v [...,3]"""
    q = np.asarray(q, F)
    v = np.asarray(v, F)
    qx, qy, qz, qw = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    ux = F(2) * (qy * v[..., 2] - qz * v[..., 1])
    uy = F(2) * (qz * v[..., 0] - qx * v[..., 2])
    uz = F(2) * (qx * v[..., 1] - qy * v[..., 0])
    out = np.stack([
        v[..., 0] + qw * ux + (qy * uz - qz * uy),
        v[..., 1] + qw * uy + (qz * ux - qx * uz),
        v[..., 2] + qw * uz + (qx * uy - qy * ux)], -1)
    return out.astype(F)


def quat_mul(a, b):
    a = np.asarray(a, F)
    b = np.asarray(b, F)
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz], -1).astype(F)


def quat_conj(q):
    q = np.asarray(q, F)
    return q * np.array([-1, -1, -1, 1], F)


def rel_pose(pi, pj):
    """Gij = Gj * Gi^-1, droid_kernels.cu:96-107.  pi, pj [...,7] -> (t [...,3], q [...,4])"""
    pi = np.asarray(pi, F)
    pj = np.asarray(pj, F)
    ti, qi = pi[..., :3], pi[..., 3:]
    tj, qj = pj[..., :3], pj[..., 3:]
    qij = quat_mul(qj, quat_conj(qi))
    tij = tj - quat_rotate(qij, ti)
    return tij.astype(F), qij


def act(t, q, X):
    """actSE3 on homogeneous points X [...,4], droid_kernels.cu:70-77"""
    X = np.asarray(X, F)
    Y = quat_rotate(q, X[..., :3]) + X[..., 3:4] * t
    return np.concatenate([Y, X[..., 3:4]], -1).astype(F)


def adjT(t, q, X):
    """adjSE3, droid_kernels.cu:79-94.  X [...,6] -> Y [...,6]"""
    X = np.asarray(X, F)
    qi = quat_conj(q)
    a = quat_rotate(qi, X[..., :3])
    b = quat_rotate(qi, X[..., 3:])
    t = np.asarray(t, F)
    u = np.stack([
        t[..., 2] * X[..., 1] - t[..., 1] * X[..., 2],
        t[..., 0] * X[..., 2] - t[..., 2] * X[..., 0],
        t[..., 1] * X[..., 0] - t[..., 0] * X[..., 1]], -1)
    v = quat_rotate(qi, u)
    return np.concatenate([a, b + v], -1).astype(F)


def so3_exp(phi):
    """expSO3, droid_kernels.cu:110-132"""
    phi = np.asarray(phi, F)
    th2 = (phi * phi).sum(-1)
    th4 = th2 * th2
    th = np.sqrt(th2)
    small = th2 < F(1e-8)
    safe = np.where(small, F(1), th)
    imag = np.where(small, F(0.5) - F(1.0 / 48.0) * th2 + F(1.0 / 3840.0) * th4,
                    np.sin(F(0.5) * safe) / safe)
    real = np.where(small, F(1) - F(1.0 / 8.0) * th2 + F(1.0 / 384.0) * th4, np.cos(F(0.5) * safe))
    return np.concatenate([imag[..., None] * phi, real[..., None]], -1).astype(F)


def se3_exp(xi):
    """expSE3, droid_kernels.cu:147-175 -> (t, q)"""
    xi = np.asarray(xi, F)
    tau, phi = xi[..., :3], xi[..., 3:]
    q = so3_exp(phi)
    th2 = (phi * phi).sum(-1)
    th = np.sqrt(th2)
    big = th > F(1e-4)
    s2 = np.where(big, th2, F(1))
    s1 = np.where(big, th, F(1))
    a = np.where(big, (F(1) - np.cos(s1)) / s2, F(0))
    b = np.where(big, (s1 - np.sin(s1)) / (s1 * s2), F(0))
    c1 = np.cross(phi, tau)
    c2 = np.cross(phi, c1)
    t = tau + a[..., None] * c1
    t = t + b[..., None] * c2
    return t.astype(F), q


def retract(xi, pose):
    """retrSE3, droid_kernels.cu:877-895: exp(xi) * G"""
    pose = np.asarray(pose, F)
    dt, dq = se3_exp(xi)
    q1 = quat_mul(dq, pose[..., 3:])
    t1 = quat_rotate(dq, pose[..., :3]) + dt
    return np.concatenate([t1, q1], -1).astype(F)


def matrix(pose):
    """4x4 matrix of a pose (float64, for identities in tests)"""
    pose = np.asarray(pose, np.float64)
    x, y, z, w = pose[3:]
    R = np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = pose[:3]
    return T
