"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement (numpy) of the native dense bundle adjustment `ba_cuda`
(/root/reference/src/lib/droid_kernels.cu:1314-1437) in the reference's own formulation,
deliberately NOT using the adjoint shortcut of the HIP kernels so the two derivations check
each other:

  per_edge_terms     projective_transform_kernel            :176-424
  accum              accum_cuda / accum_kernel              :854-874, 948-998
  schur              schur_block + EEt6x6 + Ev6x1           :1001-1093, 1222-1311
  solve              SparseBlock::solve (fp64 LLT, zero on failure)  :1192-1213
  back-substitution  EvT6x1 (rows with pose index <= 0 skipped)      :1095-1115
  retraction         pose_retr / disp_retr                  :898-946

Parity status: "parity unpinned" against the native build (nvcc, Eigen and lietorch are all
absent from this environment and the reference ships no tests).  Pinned instead by
  * finite-difference checks of residual Jacobians (tests/test_oracle_ba.py),
  * the Gauss-Newton property (one step on noise-free targets leaves the state fixed; a
    step on perturbed state reduces the weighted reprojection cost),
  * agreement of the Schur-eliminated step with the solution of the full (pose+depth)
    normal equations solved densely in float64.
"""
import numpy as np

from . import se3

F = np.float32
MIN_DEPTH = F(0.25)


def per_edge_terms(poses, disps, intr, target, weight, i, j):
    """One block of projective_transform_kernel for edge (i -> j).
    target/weight [2,h,w].  Returns dict with Hs (4x6x6), vs (2x6), Eii/Eij [6,HW], Cii, bz [HW]"""
    fx, fy, cx, cy = [F(v) for v in intr[:4]]
    _, h, w = disps.shape
    HW = h * w
    y, x = np.meshgrid(np.arange(h, dtype=F), np.arange(w, dtype=F), indexing="ij")
    u = x.reshape(-1)
    v = y.reshape(-1)
    if i == j:
        tij = np.array([-0.1, 0, 0], F)
        qij = np.array([0, 0, 0, 1], F)
    else:
        tij, qij = se3.rel_pose(poses[i], poses[j])
    Xi = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u), disps[i].reshape(-1)], -1).astype(F)
    Xj = se3.act(tij, qij, Xi)
    xx, yy, hh = Xj[:, 0], Xj[:, 1], Xj[:, 3]
    near = Xj[:, 2] < MIN_DEPTH
    with np.errstate(divide="ignore"):
        d = np.where(near, F(0), F(1) / np.where(near, F(1), Xj[:, 2])).astype(F)
    d2 = d * d
    wu = np.where(near, F(0), F(0.001) * weight[0].reshape(-1)).astype(F)
    wv = np.where(near, F(0), F(0.001) * weight[1].reshape(-1)).astype(F)
    ru = target[0].reshape(-1) - (fx * d * xx + cx)
    rv = target[1].reshape(-1) - (fy * d * yy + cy)
    o = np.zeros_like(d)

    Ju = np.stack([fx * (hh * d), o, fx * (-xx * hh * d2), fx * (-xx * yy * d2),
                   fx * (1 + xx * xx * d2), fx * (-yy * d)], -1).astype(F)
    Jzu = fx * (tij[0] * d - tij[2] * (xx * d2))
    Jv = np.stack([o, fy * (hh * d), fy * (-yy * hh * d2), fy * (-1 - yy * yy * d2),
                   fy * (xx * yy * d2), fy * (xx * d)], -1).astype(F)
    Jzv = fy * (tij[1] * d - tij[2] * (yy * d2))

    Cii = wu * Jzu * Jzu + wv * Jzv * Jzv
    bz = wu * ru * Jzu + wv * rv * Jzv
    if i == j:
        wu = np.zeros_like(wu)
        wv = np.zeros_like(wv)
    Jiu = -se3.adjT(tij, qij, Ju)
    Jiv = -se3.adjT(tij, qij, Jv)
    Jxu = np.concatenate([Jiu, Ju], -1).astype(np.float64)   # [HW,12]  (Ji | Jj)
    Jxv = np.concatenate([Jiv, Jv], -1).astype(np.float64)
    H12 = (Jxu * wu[:, None].astype(np.float64)).T @ Jxu + (Jxv * wv[:, None].astype(np.float64)).T @ Jxv
    v12 = (Jxu * (wu * ru)[:, None].astype(np.float64)).sum(0) + (Jxv * (wv * rv)[:, None].astype(np.float64)).sum(0)
    Hs = np.stack([H12[:6, :6], H12[:6, 6:], H12[6:, :6], H12[6:, 6:]]).astype(F)
    vs = np.stack([v12[:6], v12[6:]]).astype(F)
    Eii = (wu * Jzu)[None] * Jiu.T + (wv * Jzv)[None] * Jiv.T
    Eij = (wu * Jzu)[None] * Ju.T + (wv * Jzv)[None] * Jv.T
    return dict(Hs=Hs, vs=vs, Eii=Eii.astype(F), Eij=Eij.astype(F), Cii=Cii.astype(F), bz=bz.astype(F),
                r=np.stack([ru, rv]), w=np.stack([wu, wv]))


def accum(data, ix, jx):
    """out[j] = sum_{n: ix[n]==jx[j]} data[n]   (accum_cuda)"""
    out = np.zeros((len(jx),) + data.shape[1:], F)
    for j, f in enumerate(jx):
        sel = [n for n in range(len(ix)) if ix[n] == f]
        if sel:
            out[j] = data[sel].astype(np.float64).sum(0)
    return out


def ba(poses, disps, intr, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
       motion_only=False, depth_only=False, disps_sens=None):
    """In-place update of copies of poses [B,7], disps [B,h,w]; returns (poses, disps, dx, dz, info)."""
    poses = np.array(poses, F)
    disps = np.array(disps, F)
    ii = [int(v) for v in ii]
    jj = [int(v) for v in jj]
    N = len(ii)
    B, h, w = disps.shape
    HW = h * w
    P = t1 - t0
    ts = list(range(t0, t1))
    ii_exp = ts + ii
    jj_exp = ts + jj
    kx = sorted(set(ii_exp))
    kk_exp = [kx.index(f) for f in ii_exp]
    M = len(kx)
    if not motion_only:
        eta = np.asarray(eta, F).reshape(M, HW)
    dx = np.zeros((P, 6), F)
    dz = np.zeros((M, HW), F)
    info = dict(failed=0)
    for _ in range(iterations):
        terms = [per_edge_terms(poses, disps, intr, targets[n], weights[n], ii[n], jj[n]) for n in range(N)]
        Hs = np.stack([t["Hs"] for t in terms], 1)       # [4,N,6,6]
        vs = np.stack([t["vs"] for t in terms], 1)       # [2,N,6]
        Eii = np.stack([t["Eii"] for t in terms])        # [N,6,HW]
        Eij = np.stack([t["Eij"] for t in terms])
        Cii = np.stack([t["Cii"] for t in terms])
        wi = np.stack([t["bz"] for t in terms])

        # pose-pose block (SparseBlock::update_lhs / update_rhs, only i,j >= 0 checked)
        A = np.zeros((6 * P, 6 * P))
        b = np.zeros(6 * P)
        rows = ii + ii + jj + jj
        cols = ii + jj + ii + jj
        Hall = Hs.reshape(-1, 6, 6)
        for n in range(4 * N):
            i, j = rows[n] - t0, cols[n] - t0
            if 0 <= i < P and 0 <= j < P:
                A[6 * i:6 * i + 6, 6 * j:6 * j + 6] += Hall[n].astype(np.float64)
        vall = vs.reshape(-1, 6)
        for n, f in enumerate(ii + jj):
            i = f - t0
            if 0 <= i < P:
                b[6 * i:6 * i + 6] += vall[n].astype(np.float64)

        def solve(Amat, bvec):
            L = Amat.copy()
            dg = np.diag(L).copy()
            L[np.diag_indices_from(L)] = dg + ep + lm * dg
            try:
                c = np.linalg.cholesky(L)
            except np.linalg.LinAlgError:
                info["failed"] += 1
                return np.zeros((P, 6), F)
            yv = np.linalg.solve(c, bvec)
            xv = np.linalg.solve(c.T, yv)
            return xv.reshape(P, 6).astype(F)

        if motion_only:
            dx = solve(A, b)
            for p in range(P):
                poses[t0 + p] = se3.retract(dx[p], poses[t0 + p])
            continue

        if disps_sens is not None:
            m = (np.asarray(disps_sens, F)[kx] > 0).astype(F).reshape(M, HW)
            alpha = F(0.05)
            C = accum(Cii, ii, kx) + m * alpha + (1 - m) * eta
            wv = accum(wi, ii, kx) - m * alpha * (disps[kx] - np.asarray(disps_sens, F)[kx]).reshape(M, HW)
        else:
            C = accum(Cii, ii, kx) + eta
            wv = accum(wi, ii, kx)
        Q = (F(1) / C).astype(F)
        Ei = accum(Eii.reshape(N, -1), ii, ts).reshape(P, 6, HW)
        E = np.concatenate([Ei, Eij], 0)                 # rows follow ii_exp / jj_exp

        # schur_block: rows grouped by pose (j in [t0,t1)), pairs sharing the depth frame
        S = np.zeros((6 * P, 6 * P))
        sb = np.zeros(6 * P)
        rows_of_pose = [[] for _ in range(P)]
        for r in range(len(ii_exp)):
            j = jj_exp[r]
            if t0 <= j < t1:
                rows_of_pose[j - t0].append(r)
        # droid_kernels.cu:1257-1272 enumerates (pose row, pose column, rows sharing a depth frame); the same triplets are
        # visited here grouped by depth frame first (the fp64 sum of the fp32-rounded blocks only changes its order):
        # the literal four nested loops are O(P^2 deg^2) Python iterations - minutes at 128 keyframes
        rows_of_frame = [[] for _ in range(M)]
        for r in range(len(ii_exp)):
            if t0 <= jj_exp[r] < t1:
                rows_of_frame[kk_exp[r]].append(r)
        for k in range(M):
            for ra in rows_of_frame[k]:
                Ea = (E[ra] * Q[k][None]).astype(np.float64)
                pi = jj_exp[ra] - t0
                for rb in rows_of_frame[k]:
                    pj = jj_exp[rb] - t0
                    blk = (Ea @ E[rb].astype(np.float64).T).astype(F)
                    S[6 * pi:6 * pi + 6, 6 * pj:6 * pj + 6] += blk.astype(np.float64)
        for r in range(len(ii_exp)):
            i = jj_exp[r] - t0
            if 0 <= i < P:
                k = kk_exp[r]
                vv = (E[r].astype(np.float64) * (Q[k] * wv[k]).astype(np.float64)[None]).sum(1).astype(F)
                sb[6 * i:6 * i + 6] += vv.astype(np.float64)

        dx = solve(A - S, b - sb)

        # EvT6x1 with the "<= 0" skip, then dz = Q * (w - accum(dw))
        dw = np.zeros((len(ii_exp), HW), F)
        for r in range(len(ii_exp)):
            ixp = jj_exp[r] - t0
            if ixp <= 0 or ixp >= P:
                continue
            dw[r] = (E[r] * dx[ixp][:, None]).sum(0)
        dz = (Q * (wv - accum(dw, ii_exp, kx))).astype(F)
        if not depth_only:
            for p in range(P):
                poses[t0 + p] = se3.retract(dx[p], poses[t0 + p])
        for m_, f in enumerate(kx):
            disps[f] = disps[f] + dz[m_].reshape(h, w)
    return poses, disps, dx, dz, info


def reprojection_cost(poses, disps, intr, targets, weights, ii, jj):
    """sum of w * r^2 over all edges (for the GN-descent property test)"""
    c = 0.0
    for n in range(len(ii)):
        t = per_edge_terms(poses, disps, intr, targets[n], weights[n], int(ii[n]), int(jj[n]))
        c += float((t["w"].astype(np.float64) * t["r"].astype(np.float64) ** 2).sum())
    return c


def reduced_system(poses, disps, intr, targets, weights, eta_by_frame, ii, jj, t0, t1):
    """Dense reduced camera system (H = A - S, v = b - sb) BEFORE damping, for an arbitrary
    subset of edges -- used to check that per-shard systems sum to the unsharded one
    (edges partitioned by source frame).  eta_by_frame: [B,HW]."""
    poses = np.asarray(poses, F)
    disps = np.asarray(disps, F)
    ii = [int(v) for v in ii]
    jj = [int(v) for v in jj]
    N = len(ii)
    P = t1 - t0
    B, h, w = disps.shape
    HW = h * w
    A = np.zeros((6 * P, 6 * P))
    b = np.zeros(6 * P)
    if N == 0:
        return A, b
    terms = [per_edge_terms(poses, disps, intr, targets[n], weights[n], ii[n], jj[n]) for n in range(N)]
    for n in range(N):
        i, j = ii[n] - t0, jj[n] - t0
        Hs, vs = terms[n]["Hs"].astype(np.float64), terms[n]["vs"].astype(np.float64)
        for (pa, pb, blk) in ((i, i, Hs[0]), (i, j, Hs[1]), (j, i, Hs[2]), (j, j, Hs[3])):
            if 0 <= pa < P and 0 <= pb < P:
                A[6 * pa:6 * pa + 6, 6 * pb:6 * pb + 6] += blk
        if 0 <= i < P:
            b[6 * i:6 * i + 6] += vs[0]
        if 0 <= j < P:
            b[6 * j:6 * j + 6] += vs[1]
    # rows of the Schur system: self rows (pose k, depth k) for owned source frames in [t0,t1)
    # and edge rows (pose jj, depth ii)
    frames = sorted(set(ii))
    for k in frames:
        sel = [n for n in range(N) if ii[n] == k]
        C = sum(terms[n]["Cii"].astype(np.float64) for n in sel) + np.asarray(eta_by_frame[k], np.float64).reshape(-1)
        wv = sum(terms[n]["bz"].astype(np.float64) for n in sel)
        Q = 1.0 / C
        rows = []
        if t0 <= k < t1:
            rows.append((k - t0, sum(terms[n]["Eii"].astype(np.float64) for n in sel)))
        for n in sel:
            if t0 <= jj[n] < t1:
                rows.append((jj[n] - t0, terms[n]["Eij"].astype(np.float64)))
        for (pa, Ea) in rows:
            b[6 * pa:6 * pa + 6] -= (Ea * (Q * wv)[None]).sum(1)
            for (pb, Eb) in rows:
                A[6 * pa:6 * pa + 6, 6 * pb:6 * pb + 6] -= (Ea * Q[None]) @ Eb.T
    return A, b
