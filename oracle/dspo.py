"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement (numpy) of DSPO stage 2 in the reference's DENSE formulation:
  BA_with_scale_shift   /root/reference/src/geom/ba.py:127-216
  schur_solve           /root/reference/src/geom/chol.py:58-85
  projective_transform(jacobian=True) -- only Jz is used --  src/geom/projective_ops.py:96-125

`schur_solve` here is pinned by the golden fixture tests/golden/schur_solve.npz (outputs of the
importable reference function, incl. the non-PD -> zero-update branch).  The projective part
depends on lietorch (absent) -> "parity unpinned"; it shares oracle/se3.py, which is pinned by
identities.
"""
import numpy as np

from . import se3

F = np.float32


def schur_solve(H, E, C, v, w, ep=0.1, lm=1e-4, dtype=np.float32):
    """H [P,P,D,D], E [P,M,D,HW], C [M,HW], v [P,D], w [M,HW] -> dx [P,D], dz [M,HW] (float32 math
    like torch; Cholesky failure -> dx = 0, chol.py:10-17).  dtype=np.float64: the same algebra evaluated in double
    (the yardstick for how much of a difference is the fp32 evaluation's own rounding, tests/test_gpu_dspo.py)"""
    F = dtype
    P, M, D, HW = E.shape
    Hm = np.asarray(H, F).transpose(0, 2, 1, 3).reshape(P * D, P * D)
    Em = np.asarray(E, F).transpose(0, 2, 1, 3).reshape(P * D, M * HW)
    Q = (F(1) / np.asarray(C, F)).reshape(M * HW, 1)
    I = np.eye(P * D, dtype=F)
    Hm = Hm + (F(ep) + F(lm) * Hm) * I
    vv = np.asarray(v, F).reshape(P * D, 1)
    ww = np.asarray(w, F).reshape(M * HW, 1)
    S = Hm - Em @ (Q * Em.T)
    vv = vv - Em @ (Q * ww)
    try:
        Lc = np.linalg.cholesky(S.astype(np.float64))
        dx = np.linalg.solve(Lc.T, np.linalg.solve(Lc, vv.astype(np.float64))).astype(F)
    except np.linalg.LinAlgError:
        dx = np.zeros_like(vv)
    dz = Q * (ww - Em.T @ dx)
    return dx.reshape(P, D), dz.reshape(M, HW)


def block_solve(H, b, ep=0.1, lm=1e-4):
    P, _, D, _ = H.shape
    # chol.py:44-48: `I = eye(D)` broadcasts over ALL blocks, i.e. the damping term is added to
    # the diagonal of every D x D block, off-diagonal blocks included (reference quirk)
    Hb = np.asarray(H, F)
    Hb = Hb + (F(ep) + F(lm) * Hb) * np.eye(D, dtype=F)
    Hm = Hb.transpose(0, 2, 1, 3).reshape(P * D, P * D)
    try:
        Lc = np.linalg.cholesky(Hm.astype(np.float64))
        x = np.linalg.solve(Lc.T, np.linalg.solve(Lc, np.asarray(b, np.float64).reshape(-1, 1)))
    except np.linalg.LinAlgError:
        x = np.zeros((P * D, 1))
    return x.reshape(P, D).astype(F)


def ba_with_scale_shift(target, weight, eta, poses, disps, intr, ii, jj, mono, scales, shifts,
                        vmask, lm=1e-4, ep=0.1, alpha=0.01, dtype=np.float32):
    """One call of BA_with_scale_shift.  target/weight [N,h,w,2]; eta [M,h,w]; returns
    (disps, scales, shifts, dz) with the dense M x M system of the reference.  dtype=np.float64 evaluates the same
    formulation in double from the same fp32 inputs (the relative pose of an edge stays oracle/se3.py's float32)."""
    F = dtype
    poses = np.asarray(poses, F)
    disps = np.array(disps, F)
    scales = np.array(scales, F)
    shifts = np.array(shifts, F)
    ii = [int(v) for v in ii]
    jj = [int(v) for v in jj]
    N = len(ii)
    B, h, w = disps.shape
    HW = h * w
    kx = sorted(set(ii))
    kk = [kx.index(f) for f in ii]
    M = len(kx)
    y, x = np.meshgrid(np.arange(h, dtype=F), np.arange(w, dtype=F), indexing="ij")
    sa = F(np.sqrt(F(alpha)))
    Ck = np.zeros((N, HW), F)
    wk = np.zeros((N, HW), F)
    for n in range(N):
        i, j = ii[n], jj[n]
        fxi, fyi, cxi, cyi = intr[i]
        fxj, fyj, cxj, cyj = intr[j]
        X0 = np.stack([(x - cxi) / fxi, (y - cyi) / fyi, np.ones_like(x), disps[i]], -1).astype(F)
        if i == j:
            t, q = np.array([-0.1, 0, 0], F), np.array([0, 0, 0, 1], F)
        else:
            t, q = se3.rel_pose(poses[i], poses[j])
        X1 = se3.act(t, q, X0)
        Z = np.where(X1[..., 2] < F(0.1), F(1), X1[..., 2])
        d = F(1) / Z
        cu = fxj * (X1[..., 0] * d) + cxj
        cv = fyj * (X1[..., 1] * d) + cyj
        valid = ((X1[..., 2] > F(0.2)) & (X0[..., 2] > F(0.2))).astype(F)
        Jzu = fxj * d * t[0] - fxj * X1[..., 0] * d * d * t[2]
        Jzv = fyj * d * t[1] - fyj * X1[..., 1] * d * d * t[2]
        wu = F(0.001) * valid * weight[n, ..., 0]
        wv = F(0.001) * valid * weight[n, ..., 1]
        ru = target[n, ..., 0] - cu
        rv = target[n, ..., 1] - cv
        wk[n] = (-wu * ru * Jzu - wv * rv * Jzv).reshape(-1)
        Ck[n] = (wu * Jzu * Jzu + wv * Jzv * Jzv).reshape(-1)
    m = np.asarray(mono, F)[kx].reshape(M, HW)
    invalid = m < F(1e-6)
    vd = np.asarray(vmask)[kx].reshape(M, HW).astype(bool)
    sap = np.where(vd, sa * F(10), sa).astype(F)
    r_depth = sa * (disps[kx].reshape(M, HW) - (scales[kx][:, None] * m + shifts[kx][:, None]))
    J_d = sap.copy()
    J_s = (-m * sap).astype(F)
    J_q = (-sap).astype(F)
    J_d[invalid & vd] = 0
    J_s[invalid] = 0
    J_q[invalid] = 0
    J_wq = np.stack([J_s, J_q], -1)                       # [M,HW,2]
    H_wq = np.einsum("mpa,mpb->mab", J_wq, J_wq)          # [M,2,2]
    u = -np.einsum("mpa,mp->ma", J_wq, r_depth)           # [M,2]
    E_d = (J_wq * J_d[..., None]).transpose(0, 2, 1)      # [M,2,HW]
    C_proj = np.zeros((M, HW), F)
    w_proj = np.zeros((M, HW), F)
    for n in range(N):
        C_proj[kk[n]] += Ck[n]
        w_proj[kk[n]] += wk[n]
    C = C_proj + J_d * J_d + np.asarray(eta, F).reshape(M, HW)
    wv_ = -w_proj - J_d * r_depth
    Hd = np.zeros((M, M, 2, 2), F)
    Ed = np.zeros((M, M, 2, HW), F)
    for k in range(M):
        Hd[k, k] = H_wq[k]
        Ed[k, k] = E_d[k]
    dwq, dz = schur_solve(Hd, Ed, C, u, wv_, ep, lm, dtype=F)
    for k, f in enumerate(kx):
        disps[f] = disps[f] + dz[k].reshape(h, w)
        scales[f] += dwq[k, 0]
        shifts[f] += dwq[k, 1]
    disps = np.maximum(disps, F(0))
    return disps, scales, shifts, dz
