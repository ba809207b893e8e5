"""Parity of the device-resident bundle adjustment with the oracle restatement of ba_cuda."""
import numpy as np
import pytest
import torch

import glorie_slam_amd.synth as synth
from oracle import ba as oba, geom as ogeom, se3

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4   # poses 1e-4
DISP_TOL = 2e-4


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def make_problem(K, h, w, radius=2, noise=0.5, perturb=True, seed=7, extra_edges=()):
    g = synth.keyframe_graph(K=K, h=h, w=w, radius=radius, seed=seed, noise_px=noise)
    ii = np.concatenate([g["ii"], [e[0] for e in extra_edges]]).astype(np.int64)
    jj = np.concatenate([g["jj"], [e[1] for e in extra_edges]]).astype(np.int64)
    rng = np.random.default_rng(seed)
    N = len(ii)
    coords, _ = ogeom.reproject(g["poses"], g["disps"], g["intrinsics"], ii, jj)
    noise_arr = rng.normal(0, noise, (N, 2, h, w)).astype(np.float32)
    g["target"] = (coords.transpose(0, 3, 1, 2) + noise_arr).astype(np.float32)
    g["weight"] = rng.uniform(0, 1, (N, 2, h, w)).astype(np.float32)
    g["ii"], g["jj"] = ii, jj
    if perturb:
        for k in range(1, K):
            g["poses"][k] = se3.retract((rng.standard_normal(6) * 0.005).astype(np.float32), g["poses"][k])
        g["disps"] = (g["disps"] * (1 + 0.02 * rng.standard_normal(g["disps"].shape))).astype(np.float32)
    return g


def run_gpu(g, dev, t0, t1, iters, lm=1e-4, ep=0.1, motion_only=False, depth_only=False, eta=None):
    from glorie_slam_amd import droid_backends as db
    poses = _t(g["poses"], dev)
    disps = _t(g["disps"], dev)
    eta_t = _t(g["eta"] if eta is None else eta, dev)
    dx, dz = db.ba(poses, disps, _t(g["intrinsics"][0], dev), None, _t(g["target"], dev),
                   _t(g["weight"], dev), eta_t, _t(g["ii"], dev), _t(g["jj"], dev), t0, t1, iters,
                   lm, ep, motion_only, depth_only)
    torch.cuda.synchronize()
    from glorie_slam_amd import _lib
    st = _lib.default_context().ba_status()
    return poses.cpu().numpy(), disps.cpu().numpy(), dx.cpu().numpy(), dz.cpu().numpy(), st


@pytest.mark.parametrize("K,h,w,iters,radius", [(4, 12, 16, 1, 2), (5, 24, 32, 2, 2), (8, 30, 40, 2, 2),
                                                (8, 60, 80, 2, 3)])          # last: BASELINE graph G8
def test_ba_matches_oracle(gpu, K, h, w, iters, radius):
    g = make_problem(K, h, w, radius=radius)
    t0, t1 = 1, K
    rp, rd, rdx, rdz, info = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                                    g["eta"], g["ii"], g["jj"], t0, t1, iters, 1e-4, 0.1)
    assert info["failed"] == 0
    p, d, dx, dz, st = run_gpu(g, gpu, t0, t1, iters)
    assert st[0] == 0 and st[1] == K
    np.testing.assert_allclose(dx, rdx, rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(p, rp, atol=POSE_TOL)
    np.testing.assert_allclose(d, rd, atol=DISP_TOL)
    np.testing.assert_allclose(dz, rdz, rtol=5e-3, atol=2e-5)
    assert np.array_equal(p[0], g["poses"][0])  # pose 0 is fixed (t0 = 1)


def test_ba_hwc_target_layout_is_bit_identical(gpu):
    """GLORIE_BA_TARGETS_HWC: targets / weights as [N,h,w,2] (FactorGraph's layout) == the binding's [N,2,h,w]"""
    from glorie_slam_amd import droid_backends as db
    g = make_problem(6, 12, 16)
    outs = []
    for hwc in (False, True):
        poses, disps = _t(g["poses"], gpu), _t(g["disps"], gpu)
        tg, wg = _t(g["target"], gpu), _t(g["weight"], gpu)
        if hwc:
            tg, wg = tg.permute(0, 2, 3, 1).contiguous(), wg.permute(0, 2, 3, 1).contiguous()
        dx, dz = db.ba(poses, disps, _t(g["intrinsics"][0], gpu), None, tg, wg, _t(g["eta"], gpu),
                       _t(g["ii"], gpu), _t(g["jj"], gpu), 1, 6, 2, 1e-4, 0.1, False, False, targets_hwc=hwc)
        outs.append((poses.clone(), disps.clone(), dx.clone(), dz.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# one problem per solver class (tools/ba_repeat.py:SIZES): n6 = 30 / 42 dense band <4>, 60 dense band <9>, 66 / 174 sliding
# window band, 84 band <9>, 78 dense fused (bw >= 64), 474 fused (band does not fit LDS), 594 multi-kernel
@pytest.mark.parametrize("K", [6, 8, 11, 12, 14, 15, 30, 80, 100])
def test_ba_is_bitwise_reproducible(gpu, K):
    """Contract: identical inputs -> identical bits, for the reduced system (exact integer accumulators
    instead of fp64 atomics), for the solvers on a fixed system, and for glorie_ba as a whole - also while a second
    stream keeps the chip busy (which is what exposed the read/write hazard of band_eliminate in round 4: 157 of
    18000 repetitions differed by up to 9e-4 in a pose, profiles/r05_ba_repeat_before.txt).  The reference is
    deterministic by construction (droid_kernels.cu:948-998 sorted-edge accum, :1117-1219 Eigen LLT on the host)."""
    from tools import ba_repeat
    h, w, radius = ba_repeat.SIZES[K]
    p = ba_repeat.make_problem(K, h, w, radius)
    for busy in (False, True):
        bad, worst = ba_repeat.repeat_phases(p, 150, busy=busy)
        assert bad == dict(build=0, solve=0, full=0), (busy, bad, worst)


@pytest.mark.parametrize("K", [6, 14, 100])               # band solver / fused solver / multi-kernel solver
def test_ba_device_gate(gpu, K):
    """glorie_ba_set_gate: with the gate word != 0 a BA call leaves poses, disparities and the status word untouched (every
    kernel returns at once) and does not count itself; with the word == 0 it is the ungated call, bit for bit, and counts
    once.  This is how the stage-1 fallback of a depth_scale stage (depth_video.py:290-294) is decided on the device."""
    from glorie_slam_amd import droid_backends as db, _lib
    from tools import ba_repeat
    h, w, radius = ba_repeat.SIZES[K]
    g = make_problem(K, h, w, radius=radius)
    args = lambda: (_t(g["intrinsics"][0], gpu), None, _t(g["target"], gpu), _t(g["weight"], gpu), _t(g["eta"], gpu),
                    _t(g["ii"], gpu), _t(g["jj"], gpu), 1, K, 2, 1e-4, 0.1, False, False)
    ref_p, ref_d = _t(g["poses"], gpu), _t(g["disps"], gpu)
    db.ba(ref_p, ref_d, *args(), want_updates=False)
    torch.cuda.synchronize()
    st_ref = _lib.default_context().ba_status()
    hits = torch.zeros(1, dtype=torch.int32, device=gpu)
    for flag_value in (1, 0, 7):
        flag = torch.full((1,), flag_value, dtype=torch.int32, device=gpu)
        p, d = _t(g["poses"], gpu), _t(g["disps"], gpu)
        before = int(hits.item())
        db.ba(p, d, *args(), want_updates=False, gate=(flag, hits))
        torch.cuda.synchronize()
        if flag_value == 0:
            assert torch.equal(p, ref_p) and torch.equal(d, ref_d) and int(hits.item()) == before + 1
        else:
            assert torch.equal(p, _t(g["poses"], gpu)) and torch.equal(d, _t(g["disps"], gpu))
            assert int(hits.item()) == before
        assert _lib.default_context().ba_status() == st_ref
    # the gate does not outlive the call
    p, d = _t(g["poses"], gpu), _t(g["disps"], gpu)
    db.ba(p, d, *args(), want_updates=False)
    assert torch.equal(p, ref_p)


def test_ba_window_inside_graph(gpu):
    """t0 > 1: frames below t0 are fixed but still own depth maps (kx = unique(cat(ts, ii)))"""
    K = 7
    g = make_problem(K, 16, 20, radius=3)
    t0, t1 = 3, K
    rp, rd, rdx, rdz, info = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                                    g["eta"], g["ii"], g["jj"], t0, t1, 2, 1e-4, 0.1)
    p, d, dx, dz, st = run_gpu(g, gpu, t0, t1, 2)
    assert st[0] == 0
    np.testing.assert_allclose(p, rp, atol=POSE_TOL)
    np.testing.assert_allclose(d, rd, atol=DISP_TOL)
    assert np.array_equal(p[:t0], g["poses"][:t0])


def test_ba_motion_only_and_depth_only(gpu):
    K = 5
    g = make_problem(K, 16, 20)
    rp, rd, rdx, _, _ = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                               g["eta"], g["ii"], g["jj"], 1, K, 2, 1e-4, 0.1, motion_only=True)
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 2, motion_only=True)
    np.testing.assert_allclose(p, rp, atol=POSE_TOL)
    assert np.array_equal(d, g["disps"])
    rp, rd, *_ = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                        g["eta"], g["ii"], g["jj"], 1, K, 2, 1e-4, 0.1, depth_only=True)
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 2, depth_only=True)
    assert np.array_equal(p, g["poses"])
    np.testing.assert_allclose(d, rd, atol=DISP_TOL)


def test_ba_stereo_edge_and_unordered_edges(gpu):
    """ii == jj edges use the fixed baseline and only feed C, w; edge order must not matter"""
    K = 5
    g = make_problem(K, 12, 16, extra_edges=[(2, 2), (3, 3)])
    rp, rd, *_ = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                        g["eta"], g["ii"], g["jj"], 1, K, 1, 1e-4, 0.1)
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 1)
    np.testing.assert_allclose(p, rp, atol=POSE_TOL)
    np.testing.assert_allclose(d, rd, atol=DISP_TOL)
    perm = np.random.default_rng(0).permutation(len(g["ii"]))
    g2 = dict(g)
    for k in ("ii", "jj", "target", "weight"):
        g2[k] = g[k][perm]
    p2, d2, *_ = run_gpu(g2, gpu, 1, K, 1)
    np.testing.assert_allclose(p2, p, atol=2e-6)
    np.testing.assert_allclose(d2, d, atol=2e-6)


def test_ba_high_degree_frame(gpu):
    """a hub frame with more than 8 outgoing edges exercises the grouped gram path"""
    K = 14
    g = make_problem(K, 12, 16, radius=2, extra_edges=[(6, j) for j in range(K) if abs(6 - j) > 2])
    deg6 = int((g["ii"] == 6).sum())
    assert deg6 > 8
    rp, rd, *_ = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                        g["eta"], g["ii"], g["jj"], 1, K, 1, 1e-4, 0.1)
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 1)
    assert st[0] == 0
    np.testing.assert_allclose(p, rp, atol=POSE_TOL)
    np.testing.assert_allclose(d, rd, atol=DISP_TOL)


def test_ba_cholesky_failure_gives_zero_update(gpu):
    """non-PD reduced system -> zero pose update, like the reference's LLT failure branch"""
    K = 4
    g = make_problem(K, 12, 16)
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 1, lm=-3.0, ep=-1.0)
    assert st[0] & 4 and st[2] == 1
    assert np.all(dx == 0)
    assert np.array_equal(p, g["poses"])


@pytest.mark.parametrize("K", [4, 12, 92])                  # dense band / band + fused / multi-kernel solver
def test_ba_nonfinite_term_gives_zero_update(gpu, K):
    """a NaN target poisons the reduced system: zero pose update and the failure bit, as a NaN matrix does to the
    reference's LLT (droid_kernels.cu:1192-1213); the exact accumulators must not silently drop the term"""
    g = make_problem(K, 8, 12, radius=3 if K > 4 else 2)
    g["target"][1, 0, 3, 4] = np.nan
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 1)
    assert st[0] & 4 and st[0] & 32 and st[2] == 1
    assert np.all(dx == 0)
    assert np.array_equal(p, g["poses"])


def test_ba_large_window_uses_blocked_solver(gpu):
    """6P = 234 with loop closures (dense system): the one-workgroup blocked Cholesky (ba_solve_fused_kernel)"""
    g = synth.loop_graph(K=40, h=12, w=16)
    coords, _ = ogeom.reproject(g["poses"], g["disps"], g["intrinsics"], g["ii"], g["jj"])
    g["target"] = (coords.transpose(0, 3, 1, 2) + g["noise"]).astype(np.float32)
    K = g["K"]
    rp, rd, rdx, *_ = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                             g["eta"], g["ii"], g["jj"], 1, K, 1, 1e-5, 1e-2)
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 1, lm=1e-5, ep=1e-2)
    assert st[0] == 0
    np.testing.assert_allclose(dx, rdx, rtol=5e-3, atol=5e-6)
    np.testing.assert_allclose(p, rp, atol=POSE_TOL)
    np.testing.assert_allclose(d, rd, atol=DISP_TOL)


@pytest.mark.parametrize("K,extra,path", [
    (12, (), "band, 6P = 66 (just above the dense-band limit of 64)"),
    (12, ((0, 11), (11, 0), (2, 10)), "loop closure: half bandwidth >= 64 -> one-workgroup blocked Cholesky"),
    (30, (), "band, 6P = 174, several waves of window positions"),
    (30, ((1, 28), (28, 1)), "dense, 6P = 174: several 32-column blocks incl. a partial one"),
    (92, (), "6P = 546 > 540: multi-kernel blocked Cholesky in HBM"),
])
def test_ba_solver_paths_match_oracle(gpu, K, extra, path):
    """every branch of the fp64 solve (csrc/ba.hip: ba_solve_update) against the oracle's dense Cholesky"""
    g = make_problem(K, 8, 12, radius=3, extra_edges=extra)
    rp, rd, rdx, rdz, info = oba.ba(g["poses"], g["disps"], g["intrinsics"][0], g["target"], g["weight"],
                                    g["eta"], g["ii"], g["jj"], 1, K, 2, 1e-4, 0.1)
    assert info["failed"] == 0
    p, d, dx, dz, st = run_gpu(g, gpu, 1, K, 2)
    assert st[0] == 0 and st[3] == 0, path
    np.testing.assert_allclose(dx, rdx, rtol=5e-3, atol=5e-6, err_msg=path)
    np.testing.assert_allclose(p, rp, atol=POSE_TOL, err_msg=path)
    np.testing.assert_allclose(d, rd, atol=DISP_TOL, err_msg=path)


def test_ba_full_size_fixed_point_and_descent(gpu):
    """BASELINE size G8 (60x80, 36 edges): noise-free targets are a fixed point; from a
    perturbed state two GN iterations cut the reprojection cost by > 5x."""
    g = make_problem(8, 60, 80, radius=3, noise=0.0, perturb=False)
    p, d, dx, dz, st = run_gpu(g, gpu, 1, 8, 2)
    assert st[0] == 0
    assert np.abs(dx).max() < 1e-4 and np.abs(dz).max() < 1e-4
    g = make_problem(8, 60, 80, radius=3, noise=0.0, perturb=True)
    intr = g["intrinsics"][0]
    c0 = oba.reprojection_cost(g["poses"], g["disps"], intr, g["target"], g["weight"], g["ii"], g["jj"])
    p, d, *_ = run_gpu(g, gpu, 1, 8, 2)
    c1 = oba.reprojection_cost(p, d, intr, g["target"], g["weight"], g["ii"], g["jj"])
    assert c1 < 0.2 * c0


@pytest.mark.parametrize("K,h,w,world", [(7, 16, 20, 2), (38, 8, 12, 4)])
def test_ba_sharded_equals_unsharded(gpu, K, h, w, world):
    """build_system on `world` source-frame shards (one context each, on one GPU), summed like the
    RCCL all-reduce would, then solve_update on each shard == glorie_ba on the whole graph.
    K = 38: 6P = 222 unknowns, half bandwidth 41 -> the banded LDS solver, as in the multi-GPU bench."""
    from glorie_slam_amd import _lib as L, dist as gdist
    g = make_problem(K, h, w, radius=3)
    t0, t1, lm, ep = 1, K, 1e-4, 0.1
    rp, rd, rdx, rdz, st = run_gpu(g, gpu, t0, t1, 1)
    owner = gdist.shard_frames(g["ii"], world)
    B, h, w = g["disps"].shape
    n6 = 6 * (t1 - t0)
    lib = L.load()
    shards = []
    for r in range(world):
        m = gdist.local_edges(g["ii"], owner, r)
        ii_l, jj_l = g["ii"][m], g["jj"][m]
        kx = sorted(set(list(range(t0, t1)) + ii_l.tolist()))
        sh = dict(ctx=L.Context(), ii=_t(ii_l, gpu), jj=_t(jj_l, gpu), tgt=_t(g["target"][m], gpu),
                  wgt=_t(g["weight"][m], gpu), eta=_t(g["eta"][kx], gpu), M=len(kx), N=int(m.sum()),
                  poses=_t(g["poses"], gpu), disps=_t(g["disps"], gpu),
                  hv=torch.empty(n6 * n6 + n6, dtype=torch.float64, device=gpu), owned=owner == r)
        L.check(lib.glorie_ba_build_system(sh["ctx"].handle, L.ptr(sh["poses"]), L.ptr(sh["disps"]),
                                           L.ptr(_t(g["intrinsics"][0], gpu)), None, L.ptr(sh["tgt"]),
                                           L.ptr(sh["wgt"]), L.ptr(sh["eta"]), L.ptr(sh["ii"]), L.ptr(sh["jj"]),
                                           B, sh["N"], sh["M"], h, w, t0, t1, 0, L.ptr(sh["hv"]),
                                           L.stream_ptr()), "build")
        shards.append(sh)
    total = sum(sh["hv"] for sh in shards)
    disps = g["disps"].copy()
    for sh in shards:
        sh["hv"].copy_(total)
        L.check(lib.glorie_ba_solve_update(sh["ctx"].handle, L.ptr(sh["poses"]), L.ptr(sh["disps"]),
                                           L.ptr(sh["ii"]), L.ptr(sh["jj"]), B, sh["N"], sh["M"], h, w, t0, t1,
                                           lm, ep, 0, 0, L.ptr(sh["hv"]), None, None, L.stream_ptr()), "solve")
        torch.cuda.synchronize()
        np.testing.assert_allclose(sh["poses"].cpu().numpy(), rp, atol=2e-6 if K < 20 else 2e-5)
        d = sh["disps"].cpu().numpy()
        disps[:len(sh["owned"])][sh["owned"]] = d[:len(sh["owned"])][sh["owned"]]
    np.testing.assert_allclose(disps, rd, atol=2e-6 if K < 20 else 2e-5)


@pytest.mark.parametrize("P,bw", [(1, 5), (7, 41), (7, 5), (11, 41), (11, 65), (25, 41), (25, 63), (25, 64),
                                  (25, 149), (49, 41), (49, 293), (90, 41), (90, 539), (91, 41), (91, 545),
                                  (167, 200), (300, 41), (300, 700)])
def test_solver_kernels_on_synthetic_spd_systems(gpu, P, bw):
    """the fp64 solve in isolation: glorie_ba_solve_update (N = 0: no depth frames) on a random banded SPD
    system written straight into the [H | v] buffer, against numpy.  Covers the dense-band kernel (6P <= 64),
    the banded LDS kernel, the one-workgroup blocked Cholesky (half bandwidth >= 64 or band too large for LDS)
    and the multi-launch path (6P > 540)."""
    from glorie_slam_amd import _lib as L
    n = 6 * P
    rng = np.random.default_rng(100 * P + bw)
    A = rng.standard_normal((n, n))
    r, c = np.indices((n, n))
    A[np.abs(r - c) > bw] = 0.0
    H = A @ A.T * 0.05
    H[np.abs(r - c) > bw] = 0.0                       # keep the band exact (A A^T widens it)
    H = 0.5 * (H + H.T) + np.diag(np.abs(H).sum(1) + 1.0)   # diagonally dominant -> SPD
    v = rng.standard_normal(n)
    lm, ep = 1e-4, 0.1
    ref = np.linalg.solve(H + np.diag(ep + lm * np.diag(H)), v)
    lib, ctx = L.load(), L.Context()
    B = P + 1
    poses = torch.zeros(B, 7, device=gpu)
    poses[:, 6] = 1.0
    disps = torch.ones(B, 2, 2, device=gpu)
    hv = torch.zeros(n * n + n, dtype=torch.float64, device=gpu)
    L.check(lib.glorie_ba_build_system(ctx.handle, L.ptr(poses), L.ptr(disps), None, None, None, None, None,
                                       None, None, B, 0, 1, 2, 2, 1, B, 0, L.ptr(hv), L.stream_ptr()), "build")
    low = np.tril(H)                                   # the kernels read the lower triangle only
    hv.copy_(torch.from_numpy(np.concatenate([low.reshape(-1), v])))
    dx = torch.zeros(P, 6, device=gpu)
    L.check(lib.glorie_ba_solve_update(ctx.handle, L.ptr(poses), L.ptr(disps), None, None, B, 0, 1, 2, 2, 1, B,
                                       lm, ep, 0, 0, L.ptr(hv), L.ptr(dx), None, L.stream_ptr()), "solve")
    torch.cuda.synchronize()
    st = ctx.ba_status()
    assert st[0] == 0 and st[3] == 0
    got = dx.cpu().numpy().reshape(-1).astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-7 * np.abs(ref).max())


def test_large_system_failed_iteration_does_not_poison_the_next(gpu):
    """n = 546 > 540 (multi-launch Cholesky): a factorisation that fails zeroes ITS update only; the next
    solve on the same context - the second Gauss-Newton iteration of one glorie_ba call re-uses the status
    word without another ba_prepare - factorises on its own like SparseBlock::solve (droid_kernels.cu:1192-1213)"""
    from glorie_slam_amd import _lib as L
    P = 91
    n = 6 * P
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, n)) * 0.05
    H = A @ A.T + np.eye(n)
    v = rng.standard_normal(n)
    lm, ep = 1e-4, 0.1
    ref = np.linalg.solve(H + np.diag(ep + lm * np.diag(H)), v)
    lib, ctx = L.load(), L.Context()
    B = P + 1
    poses = torch.zeros(B, 7, device=gpu)
    poses[:, 6] = 1.0
    disps = torch.ones(B, 2, 2, device=gpu)
    hv = torch.zeros(n * n + n, dtype=torch.float64, device=gpu)
    L.check(lib.glorie_ba_build_system(ctx.handle, L.ptr(poses), L.ptr(disps), None, None, None, None, None,
                                       None, None, B, 0, 1, 2, 2, 1, B, 0, L.ptr(hv), L.stream_ptr()), "build")

    def solve(Hm):
        hv.copy_(torch.from_numpy(np.concatenate([np.tril(Hm).reshape(-1), v])))
        dx = torch.full((P, 6), 7.0, device=gpu)
        L.check(lib.glorie_ba_solve_update(ctx.handle, L.ptr(poses), L.ptr(disps), None, None, B, 0, 1, 2, 2, 1, B,
                                           lm, ep, 0, 0, L.ptr(hv), L.ptr(dx), None, L.stream_ptr()), "solve")
        torch.cuda.synchronize()
        return dx.cpu().numpy().reshape(-1).astype(np.float64), ctx.ba_status()

    bad = H.copy()
    bad[300, 300] = -50.0                                  # not positive definite (fails in the 10th column block)
    dx, st = solve(bad)
    assert np.all(dx == 0) and st[0] & 4 and st[2] == 1
    p_before = poses.clone()
    dx, st = solve(H)                                      # same context, no ba_prepare in between
    assert st[2] == 1                                      # no new failure
    np.testing.assert_allclose(dx, ref, rtol=2e-5, atol=1e-7 * np.abs(ref).max())
    assert not torch.equal(poses, p_before)                # ... and the update was applied
