"""Pins of the HOST side of the path against the imported reference (this container only; /root/reference never
travels): graph topology + edge bookkeeping, the call schedules of the drivers, and the call signatures slam.py /
tracker.py / mapper.py rely on.

    python tests/golden/make_pins.py

  topology.npz      FactorGraph of /root/reference/src/factor_graph.py (CPU, `MatrixVideo` answering `distance` from a
                    stored matrix) put through the scripts of recording.topology_cases(): edge lists, ages, inactive /
                    bad lists, target / weight row tags after EVERY operation (add_neighborhood_factors :312-320,
                    add_proximity_factors :323-383, add_backend_proximity_factors :386-462, add_factors incl. the
                    factor limit :95-143, rm_factors :146-170, rm_keyframe :173-209, filter_edges :68-75,
                    __filter_repeated_edges :42-53).  Equal sort keys: see mint_topology.
  driver_traces.json  Frontend (src/frontend.py:40-117), Backend (src/backend.py:27-97) and PoseTrajectoryFiller
                    (src/trajectory_filler.py:34-107) run over `RecordingGraph` / `DriverVideo`: every call they make,
                    in order, with its arguments.
  signatures.json   inspect.signature of every class / method / function of the hot-path modules.

Stand-ins used while minting (absent third-party packages, never executed arithmetic of the path): colorama, tqdm-free
printing, torchvision, src.mono_estimators, src.utils.datasets (needs cv2), `torch.Tensor.cuda` -> identity (the filler
calls `.cuda()` on index tensors, trajectory_filler.py:70-71), and for the filler ONLY a shape-only `lietorch.SE3`
(group arithmetic replaced by placeholders: the trace pins the schedule and the integer arguments, not pose values;
the interpolation itself is checked against scipy in tests/test_oracle_se3.py / test_gpu_long_graphs.py).
"""
import json
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import recording as R  # noqa: E402
from make_golden import install_stubs  # noqa: E402


def install_more_stubs():
    install_stubs()
    for name in ("faiss", "faiss.contrib", "faiss.contrib.torch_utils", "cv2", "open3d", "wandb", "torchvision",
                 "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    col = types.ModuleType("colorama")

    class _Any:
        def __getattr__(self, k):
            return ""

    col.Fore, col.Style = _Any(), _Any()
    sys.modules["colorama"] = col
    ds = types.ModuleType("src.utils.datasets")
    ds.load_mono_depth = None
    ds.get_dataset = None
    ds.BaseDataset = object
    sys.modules["src.utils.datasets"] = ds
    me = types.ModuleType("src.mono_estimators")
    me.get_mono_depth_estimator = me.predit_mono_depth = None
    sys.modules["src.mono_estimators"] = me


def mint_topology():
    """Equal keys: the reference calls torch.argsort / torch.sort without `stable=` (factor_graph.py:114,360,416) on CUDA
    tensors, where the sort is cub's radix sort and equal keys stay in index order; torch's CPU sort (this container) is
    NOT stable beyond 16 elements.  The sorts are therefore forced stable while minting, so that the fixture holds the
    order the reference produces on its own device (edge ages are integers: ties are the rule, not the exception)."""
    from src.factor_graph import FactorGraph
    arrays, meta = {}, {}
    real_argsort, real_sort = torch.argsort, torch.sort
    torch.argsort = lambda x, *a, **k: real_argsort(x, *a, **{**k, "stable": True})
    torch.sort = lambda x, *a, **k: real_sort(x, *a, **{**k, "stable": True})
    try:
        _mint_topology_cases(FactorGraph, arrays, meta)
    finally:
        torch.argsort, torch.sort = real_argsort, real_sort
    np.savez_compressed(os.path.join(OUT, "topology.npz"), meta=np.array(json.dumps(meta)), **arrays)


def _mint_topology_cases(FactorGraph, arrays, meta):
    for name, (kind, K, seed, max_factors, corr_impl, script) in R.topology_cases().items():
        d = R.distance_matrix(kind, K, seed)
        video = R.MatrixVideo(d)
        graph = FactorGraph(video, None, device="cpu", corr_impl=corr_impl, max_factors=max_factors)
        states = R.run_topology_script(graph, video, script)
        arrays["d_" + name] = d
        meta[name] = {"kind": kind, "K": K, "seed": seed, "max_factors": max_factors, "corr_impl": corr_impl,
                      "script": script, "states": states}
        last = [s for s in states if "ii" in s][-1]
        print(f"  {name}: {len(script)} operations, final edges {len(last['ii'])}, inactive {len(last['ii_inac'])}, "
              f"bad {len(last['ii_bad'])}, returns {[s['ret'] for s in states if s.get('ret') is not None]}")


def _patch(module_names):
    def patch(cls):
        for m in module_names:
            setattr(sys.modules[m], "FactorGraph", cls)
    return patch


def mint_driver_traces():
    import src.backend
    import src.frontend
    out = {"frontend": {}, "backend": {}, "filler": {}}
    patch = _patch(["src.frontend", "src.backend"])
    for name, sc in R.frontend_scenarios().items():
        out["frontend"][name] = R.normalise_events(R.run_frontend(src.frontend.Frontend, patch, sc))
    for name, sc in R.backend_scenarios().items():
        out["backend"][name] = R.normalise_events(R.run_backend(src.backend.Backend, patch, sc))

    # PoseTrajectoryFiller: shape-only SE3 (see the module docstring)
    lt = sys.modules["lietorch"]

    class SE3:
        def __init__(self, data):
            self.data = data

        def __getitem__(self, i):
            return SE3(self.data[i])

        def __mul__(self, o):
            return SE3(o.data)

        def inv(self):
            return self

        def log(self):
            return torch.zeros(self.data.shape[0], 6)

        @staticmethod
        def exp(w):
            d = torch.zeros(w.shape[0], 7)
            d[:, 6] = 1
            return SE3(d)

    lt.SE3 = SE3
    lt.cat = lambda xs, dim=0: SE3(torch.cat([x.data for x in xs], dim))
    import src.trajectory_filler as tf
    tf.SE3 = SE3
    tf.tqdm = lambda x: x
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        patch = _patch(["src.trajectory_filler"])
        for name, sc in R.filler_scenarios().items():
            out["filler"][name] = R.normalise_events(R.run_filler(tf.PoseTrajectoryFiller, patch, sc))
    finally:
        torch.Tensor.cuda = real_cuda
    with open(os.path.join(OUT, "driver_traces.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    for k, v in out.items():
        print(f"  {k}: " + ", ".join(f"{n} ({len(e)} events)" for n, e in v.items()))


# the hot-path modules of SURVEY 8(b): what slam.py:5-17, tracker.py:1-3,25-33,56-69 and mapper.py:10-19,48-53,196-680
# import and call
SIGNATURE_TARGETS = {
    "src.modules.droid_net.droid_net": ["DroidNet", "UpdateModule", "GraphAgg", "cvx_upsample", "upsample_disp"],
    "src.modules.droid_net.gru": ["ConvGRU"],
    "src.modules.droid_net.corr": ["CorrBlock", "AltCorrBlock"],
    "src.modules.droid_net.extractor": ["BasicEncoder", "ResidualBlock"],
    "src.factor_graph": ["FactorGraph"],
    "src.depth_video": ["DepthVideo"],
    "src.frontend": ["Frontend"],
    "src.backend": ["Backend"],
    "src.motion_filter": ["MotionFilter"],
    "src.trajectory_filler": ["PoseTrajectoryFiller"],
    "src.neural_point": ["NeuralPointCloud", "proj_depth_map", "update_points_pos", "get_proxy_render_depth", "get_scale"],
    "src.utils.Renderer": ["Renderer"],
    "src.modules.conv_onet.models.decoder": ["POINT", "MLP_geometry", "MLP_color", "MLP_col_neighbor",
                                             "GaussianFourierFeatureTransform"],
    "src.utils.common": ["raw2outputs_nerf_color", "get_rays", "get_rays_from_uv", "align_scale_and_shift",
                         "get_samples", "get_samples_with_pixel_grad", "select_uv", "get_sample_uv",
                         "get_sample_uv_with_grad", "get_tensor_from_camera", "get_camera_from_tensor",
                         "quad2rotation", "setup_seed", "update_cam"],
    "src.geom.projective_ops": ["coords_grid", "iproj", "proj", "actp", "projective_transform", "induced_flow",
                                "extract_intrinsics"],
    "src.geom.ba": ["BA", "BA_with_scale_shift", "MoBA"],
    "src.geom.chol": ["schur_solve", "block_solve"],
}


def mint_signatures():
    import importlib
    import inspect
    out = {}
    for mod, names in SIGNATURE_TARGETS.items():
        m = importlib.import_module(mod)
        for n in names:
            obj = getattr(m, n)
            key = f"{mod}.{n}"
            if inspect.isclass(obj):
                entry = {}
                for mn, fn in vars(obj).items():
                    if isinstance(fn, (staticmethod, classmethod)):
                        fn = fn.__func__
                    if not inspect.isfunction(fn):
                        continue
                    if mn.startswith("_") and not (mn.startswith("__") and mn.endswith("__")):
                        continue                                   # name-mangled privates (__update, __fill, ...) are internal
                    if mn.startswith(f"_{n}__"):
                        continue
                    entry[mn] = R.signature_of(fn)
                out[key] = {"kind": "class", "methods": entry}
            else:
                out[key] = {"kind": "function", "signature": R.signature_of(obj)}
    # the nine entry points of the pybind module (src/lib/droid.cpp:239-252) have no Python signature: names + arity
    out["droid_backends"] = {"kind": "pybind", "names": {
        "ba": 16, "frame_distance": 6, "projmap": 5, "depth_filter": 5, "iproj": 3, "altcorr_forward": 4,
        "altcorr_backward": 5, "corr_index_forward": 3, "corr_index_backward": 4}}
    with open(os.path.join(OUT, "signatures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    n_cls = sum(1 for v in out.values() if v["kind"] == "class")
    n_m = sum(len(v["methods"]) for v in out.values() if v["kind"] == "class")
    print(f"  signatures.json: {n_cls} classes / {n_m} methods, {sum(1 for v in out.values() if v['kind'] == 'function')} functions")


# ------------------------------------------------------------------------------------------------------------------
# lietorch stand-in (ABSENT third-party dependency, princeton-vl/lietorch v0.2) for the geometry fixtures.  Written
# from the group's definition with rotation MATRICES (not the quaternion formulas of csrc/se3.hiph, oracle/se3.py or
# glorie_slam_amd/lie.py, so that agreement is a check): data [t, q(xyzw)], tangent [tau, phi],
# T * X = (R X + t X_w, X_w) on homogeneous points, adjT(a) = Ad(T)^T a with Ad = [[R, [t]x R], [0, R]],
# retr(a) = exp(a) * T.  Only what projective_ops.py / ba.py call.
# ------------------------------------------------------------------------------------------------------------------
def _rot(q):
    x, y, z, w = q.unbind(-1)
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                        torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def _quat(R):
    """rotation matrix -> unit quaternion xyzw (w >= 0 branch is enough for the fixtures' small rotations)"""
    w = torch.sqrt(torch.clamp(1 + R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2], min=1e-12)) / 2
    x = (R[..., 2, 1] - R[..., 1, 2]) / (4 * w)
    y = (R[..., 0, 2] - R[..., 2, 0]) / (4 * w)
    z = (R[..., 1, 0] - R[..., 0, 1]) / (4 * w)
    return torch.stack([x, y, z, w], -1)


def _skew(v):
    o = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([o, -v[..., 2], v[..., 1]], -1), torch.stack([v[..., 2], o, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], o], -1)], -2)


class MatSE3:
    manifold_dim = 6

    def __init__(self, data):
        self.data = data

    @property
    def shape(self):
        return self.data.shape[:-1]

    def vec(self):
        return self.data

    def __getitem__(self, i):
        return MatSE3(self.data[i])

    def _Rt(self):
        return _rot(self.data[..., 3:]), self.data[..., :3]

    def inv(self):
        R, t = self._Rt()
        Ri = R.transpose(-1, -2)
        return MatSE3(torch.cat([-(Ri @ t.unsqueeze(-1)).squeeze(-1), _quat(Ri)], -1))

    def __mul__(self, o):
        R, t = self._Rt()
        if isinstance(o, MatSE3):
            R2, t2 = o._Rt()
            return MatSE3(torch.cat([(R @ t2.unsqueeze(-1)).squeeze(-1) + t, _quat(R @ R2)], -1))
        Y = (R @ o[..., :3].unsqueeze(-1)).squeeze(-1) + t * o[..., 3:4]
        return torch.cat([Y, o[..., 3:4]], -1)

    def adjT(self, a):
        R, t = self._Rt()
        Ad = torch.zeros(*R.shape[:-2], 6, 6, dtype=R.dtype)
        Ad[..., :3, :3] = R
        Ad[..., :3, 3:] = _skew(t) @ R
        Ad[..., 3:, 3:] = R
        return (Ad.transpose(-1, -2) @ a.unsqueeze(-1)).squeeze(-1)

    @staticmethod
    def exp(xi):
        tau, phi = xi[..., :3].double(), xi[..., 3:].double()
        th = phi.norm(dim=-1, keepdim=True).clamp(min=1e-12)[..., None]
        K = _skew(phi)
        I = torch.eye(3, dtype=torch.float64).expand(K.shape)
        R = I + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * (K @ K)
        V = I + (1 - torch.cos(th)) / th ** 2 * K + (th - torch.sin(th)) / th ** 3 * (K @ K)
        return MatSE3(torch.cat([(V @ tau.unsqueeze(-1)).squeeze(-1), _quat(R)], -1).float())

    def retr(self, a):
        return MatSE3.exp(a) * self


def geometry_scene(seed=31, K=5, h=6, w=8):
    """K keyframes on a gentle arc looking at a wavy wall 1.5 - 3 m away, 1/8-resolution intrinsics of a 64 x 48 image"""
    g = torch.Generator().manual_seed(seed)
    k = torch.arange(K, dtype=torch.float32)
    ang = 0.03 * k
    axis = torch.tensor([0.1, 1.0, 0.05]) / torch.tensor([0.1, 1.0, 0.05]).norm()
    q = torch.cat([axis[None] * torch.sin(ang / 2)[:, None], torch.cos(ang / 2)[:, None]], -1)
    t = torch.stack([0.06 * k, 0.01 * k, -0.02 * k], -1)
    poses = torch.cat([t, q], -1)
    y, x = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    disps = 1.0 / (2.0 + 0.5 * torch.sin(2 * torch.pi * x / w)[None] * torch.cos(2 * torch.pi * y / h)[None]
                   + 0.1 * k[:, None, None]) + 0.01 * torch.rand(K, h, w, generator=g)
    intr = torch.tensor([8.0, 8.5, 3.6, 2.7]).repeat(K, 1)
    ii = torch.tensor([0, 1, 1, 2, 2, 3, 3, 4, 4, 2, 0, 3])
    jj = torch.tensor([1, 0, 2, 1, 3, 2, 4, 3, 2, 4, 2, 3])          # the last edge is a stereo edge (ii == jj)
    N = ii.shape[0]
    target = torch.randn(1, N, h, w, 2, generator=g) * 0.05
    weight = torch.rand(1, N, h, w, 2, generator=g)
    eta = 0.2 * (torch.rand(K, h, w, generator=g) * 0.02 + 1e-3) + 1e-7
    scale = torch.tensor([1.3, 0.8, 1.1, 0.9, 1.6])[:K]
    shift = torch.tensor([0.03, -0.02, 0.0, 0.04, -0.03])[:K]
    mono = (disps - shift[:, None, None]) / scale[:, None, None] * (1 + 0.03 * torch.randn(K, h, w, generator=g))
    mono[torch.rand(K, h, w, generator=g) < 0.1] = 0.0
    vmask = torch.rand(K, h, w, generator=g) < 0.6
    return dict(poses=poses, disps=disps, intr=intr, ii=ii, jj=jj, target=target, weight=weight, eta=eta,
                mono=mono, vmask=vmask, scales0=scale * 1.05, shifts0=shift + 0.01)


def mint_geometry():
    """pops.npz / ba_scale_shift.npz / ba_python.npz: the reference's projective_ops.projective_transform (:96-125),
    ba.BA_with_scale_shift (:127-216) and ba.BA (:34-121) executed on CPU with MatSE3 in lietorch's place"""
    import src.geom.projective_ops as pops
    import src.geom.ba as rba
    pops.SE3 = MatSE3
    real_as_tensor = torch.as_tensor
    torch.as_tensor = lambda *a, **k: real_as_tensor(*a, **{kk: ("cpu" if kk == "device" else v) for kk, v in k.items()})
    try:
        sc = geometry_scene()
        P = MatSE3(sc["poses"][None].clone())
        coords_t = pops.projective_transform(P, sc["disps"][None], sc["intr"][None], sc["ii"], sc["jj"])
        x1, valid, (Ji, Jj, Jz) = pops.projective_transform(MatSE3(sc["poses"][None].clone()), sc["disps"][None],
                                                            sc["intr"][None], sc["ii"], sc["jj"], jacobian=True)
        # targets = reprojection + noise, like a converged update operator
        target = x1 + sc["target"]
        inputs = {k: v.numpy() for k, v in sc.items() if k != "target"}
        inputs["target"] = target.numpy()
        np.savez_compressed(os.path.join(OUT, "pops.npz"), coords=x1.numpy(), valid=valid.numpy(), Ji=Ji.numpy(),
                            Jj=Jj.numpy(), Jz=Jz.numpy(), coords_nojac=coords_t[0].numpy(), **inputs)
        print("  pops.npz: coords", tuple(x1.shape), "valid fraction", float(valid.mean()))

        # DSPO stage 2: two sequential calls exactly like depth_video.py:263-276 (alpha = 0.01); only edges whose source
        # frame is in `ii` get an eta row (eta is [M,h,w] over unique(ii))
        kx = torch.unique(sc["ii"])
        eta = sc["eta"][kx]
        poses, disps = MatSE3(sc["poses"][None].clone()), sc["disps"][None].clone()
        scales, shifts = sc["scales0"].clone(), sc["shifts0"].clone()
        out = {}
        for it in range(2):
            poses, disps, wqs = rba.BA_with_scale_shift(target, sc["weight"], eta, poses, disps, sc["intr"][None],
                                                        sc["ii"], sc["jj"], sc["mono"][None], scales[None], shifts[None],
                                                        sc["vmask"][None], 0, 1e-4, 0.1, alpha=0.01)
            scales, shifts = wqs[0, :, 0], wqs[0, :, 1]
            out[f"disps_{it}"] = disps[0].numpy().copy()
            out[f"wqs_{it}"] = wqs[0].numpy().copy()
        np.savez_compressed(os.path.join(OUT, "ba_scale_shift.npz"), eta_rows=eta.numpy(), **out)
        print("  ba_scale_shift.npz: |d disps|", float((disps[0] - sc["disps"]).abs().max()),
              "scales", [round(float(v), 4) for v in scales])

        # stage 1 in the reference's PYTHON formulation (ba.py:34-121; not what the product path runs - droid_backends.ba
        # is - but the same normal equations up to: MIN_DEPTH 0.2 vs 0.25, damping before vs after the Schur
        # complement, no `<= 0` skip): one iteration, fixedp = 1
        # without the stereo edge: ba.py keeps its pose terms, the native kernel drops them (droid_kernels.cu:307-311)
        keep = sc["ii"] != sc["jj"]
        poses, disps = rba.BA(target[:, keep], sc["weight"][:, keep], eta, MatSE3(sc["poses"][None].clone()),
                              sc["disps"][None].clone(), sc["intr"][None], sc["ii"][keep], sc["jj"][keep], None, 1e-4, 0.1,
                              fixedp=1)
        np.savez_compressed(os.path.join(OUT, "ba_python.npz"), poses=poses.data[0].numpy(), disps=disps[0].numpy())
        print("  ba_python.npz: |d t|", float((poses.data[0, :, :3] - sc["poses"][:, :3]).abs().max()),
              "|d disps|", float((disps[0] - sc["disps"]).abs().max()))
    finally:
        torch.as_tensor = real_as_tensor


def main():
    install_more_stubs()
    torch.set_num_threads(2)
    print("topology.npz")
    mint_topology()
    print("driver_traces.json")
    mint_driver_traces()
    print("signatures.json")
    mint_signatures()
    print("geometry fixtures")
    mint_geometry()


if __name__ == "__main__":
    main()
