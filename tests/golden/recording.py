"""Test harness shared by the fixture-minting script (tests/golden/make_pins.py, which drives the REFERENCE's classes in
the build container) and by the tests (which drive this repository's mirrors): fake keyframe buffers, a recording factor
graph, and the scripted scenarios both sides are put through.  Nothing here is reference code; the method names and
argument lists of the mocks are the interface the drivers call (/root/reference/src/factor_graph.py:95,146,173,212,259,
312,323,386; src/depth_video.py:58,166,363,327,146), restated so that a call made positionally on one side and by
keyword on the other binds to the same record.
"""
import contextlib
import inspect

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------------
# fake keyframe buffer for the TOPOLOGY scripts: distances come from a stored matrix
# ------------------------------------------------------------------------------------------------------------------
class Counter:
    def __init__(self, value=0):
        self.value = value


class MatrixVideo:
    """what FactorGraph needs of a DepthVideo, with `distance` answered from a stored [K, K] matrix and `reproject`
    returning a target whose value encodes the edge (100 * i + j): the bookkeeping of target / weight rows through
    rm_factors / rm_keyframe shows in the values"""

    def __init__(self, dmat, ht=16, wd=16, dim=4, device="cpu"):
        K = dmat.shape[0]
        self.dmat = torch.as_tensor(np.asarray(dmat, np.float32))
        self.device = device
        self.ht, self.wd, self.down_scale = ht, wd, 1
        self.counter = Counter(K)
        B = K + 2
        g = torch.Generator().manual_seed(5)
        self.timestamp = torch.arange(B, dtype=torch.float32) * 3.0
        self.images = torch.arange(B, dtype=torch.float32)[:, None, None, None].repeat(1, 3, 2, 2)
        self.dirty = torch.zeros(B, dtype=torch.bool)
        self.npc_dirty = torch.zeros(B, dtype=torch.bool)
        self.poses = torch.arange(B, dtype=torch.float32)[:, None].repeat(1, 7)
        self.disps = torch.ones(B, ht, wd) * torch.arange(1, B + 1, dtype=torch.float32)[:, None, None]
        self.disps_up = torch.zeros(B, 2, 2)
        self.intrinsics = torch.ones(B, 4)
        self.depth_scale = torch.arange(B, dtype=torch.float32)
        self.depth_shift = torch.arange(B, dtype=torch.float32) * 0.5
        self.mono_disps = torch.zeros(B, 2, 2)
        self.valid_depth_mask = torch.zeros(B, 2, 2, dtype=torch.bool)
        self.valid_depth_mask_small = torch.zeros(B, 2, 2, dtype=torch.bool)
        self.nets = torch.randn(B, dim, ht, wd, generator=g)
        self.inps = torch.randn(B, dim, ht, wd, generator=g)
        self.fmaps = torch.randn(B, 1, dim, ht, wd, generator=g)

    @contextlib.contextmanager
    def get_lock(self):
        yield

    def distance(self, ii=None, jj=None, beta=0.3, bidirectional=True):
        ii = torch.as_tensor(ii).long().reshape(-1).cpu()
        jj = torch.as_tensor(jj).long().reshape(-1).cpu()
        return self.dmat[ii, jj].clone().to(self.device)

    def reproject(self, ii, jj, motion=None):
        n = int(ii.shape[0])
        tag = (100 * ii + jj).float().reshape(1, n, 1, 1, 1)
        return tag.expand(1, n, self.ht, self.wd, 2).clone().to(self.device), None


def graph_state(graph, video, ret=None):
    """what a topology script records after every operation"""
    L = lambda t: [] if t is None else [int(v) for v in t.detach().cpu().reshape(-1).tolist()]
    F = lambda t: [] if t is None or t.numel() == 0 else [round(float(v), 6) for v in t.detach().cpu().float().tolist()]
    tag = lambda t: None if t is None else (t[0, :, 0, 0, 0] if t.shape[1] else t[0, :0, 0, 0, 0])
    mean = lambda t: None if t is None else (t.float().mean(dim=[0, 2, 3, 4]) if t.shape[1] else None)
    if graph.ii is None:                                   # after clear_edges
        return {"cleared": True, "ret": ret}
    st = {"ii": L(graph.ii), "jj": L(graph.jj), "age": L(graph.age),
          "ii_inac": L(graph.ii_inac), "jj_inac": L(graph.jj_inac), "ii_bad": L(graph.ii_bad), "jj_bad": L(graph.jj_bad),
          "target_tag": F(tag(graph.target)), "target_inac_tag": F(tag(graph.target_inac)),
          "weight_mean": F(mean(graph.weight)), "weight_inac_mean": F(mean(graph.weight_inac)),
          "net_rows": None if graph.net is None else int(graph.net.shape[1]),
          "counter": int(video.counter.value), "timestamp": F(video.timestamp), "ret": ret}
    return st


def edge_weight_rule(ii, jj):
    """deterministic per-edge confidence for the `weights` operation: some long edges fall below filter_edges'
    0.001, everything else gets a distinct value"""
    ii = ii.detach().cpu().double()
    jj = jj.detach().cpu().double()
    low = ((ii + 2 * jj) % 3 == 0)
    return torch.where(low, torch.full_like(ii, 0.0005), 0.5 + 0.001 * (ii * 31 + jj)).float()


def run_topology_script(graph, video, script):
    """applies the operations of a script to a FactorGraph (the reference's or the mirror) and returns the state after
    every one.  Operations are plain lists (they are stored in the fixture as JSON)."""
    states = []
    for op in script:
        kind, a = op[0], op[1:]
        ret = None
        if kind == "neigh":
            graph.add_neighborhood_factors(a[0], a[1], r=a[2])
        elif kind == "prox":
            t0, t1, rad, nms, beta, thresh, remove = a
            graph.add_proximity_factors(t0, t1, rad=rad, nms=nms, beta=beta, thresh=thresh, remove=remove)
        elif kind == "backprox":
            t_start, t_end, nms, radius, thresh, max_factors, beta, t_start_loop, loop = a
            ret = graph.add_backend_proximity_factors(t_start, t_end, nms, radius, thresh, max_factors, beta,
                                                      t_start_loop, loop)
            ret = int(ret)
        elif kind == "add":
            graph.add_factors(list(a[0]), list(a[1]), remove=a[2])
        elif kind == "age":
            graph.age += a[0]
        elif kind == "rm_age":
            graph.rm_factors(graph.age > a[0], store=a[1])
        elif kind == "rm_lt":
            graph.rm_factors(graph.ii < a[0], store=a[1])
        elif kind == "weights":
            w = edge_weight_rule(graph.ii, graph.jj).to(graph.weight.device)
            graph.weight = w.reshape(1, -1, 1, 1, 1).expand_as(graph.weight).clone()
        elif kind == "filter_edges":
            graph.filter_edges()
        elif kind == "rm_keyframe":
            graph.rm_keyframe(a[0])
            video.counter.value -= 1
        elif kind == "counter":
            video.counter.value = a[0]
        else:
            raise ValueError(kind)
        states.append(graph_state(graph, video, ret))
    return states


# ------------------------------------------------------------------------------------------------------------------
# distance matrices of the topology cases
# ------------------------------------------------------------------------------------------------------------------
def distance_matrix(kind, K, seed):
    """[K, K] float32 symmetric 'mean flow' matrices: a smooth trajectory (distance grows with |i - j|), quantised to
    half pixels so that EXACT TIES exist, with values above the 100 cut-off, infinities, and - for 'loop' - a revisit
    of the first frames at the end of the sequence (small distances between indices more than 20 apart).
    Ties: the reference orders candidates with torch.argsort / torch.sort without `stable=`; on its device (CUDA) that
    is cub's radix sort, which keeps equal keys in index order, while torch's CPU sort does not (n > 16).  The fixture
    is minted with the sorts forced stable (make_pins.py) = the order the reference produces on the GPU; the build
    sorts with stable=True / numpy kind="stable" explicitly."""
    rng = np.random.default_rng(seed)
    i, j = np.meshgrid(np.arange(K), np.arange(K), indexing="ij")
    gap = np.abs(i - j).astype(np.float64)
    d = 4.0 * gap + rng.uniform(0, 6, (K, K))
    if kind == "loop":
        pos = np.arange(K, dtype=np.float64)
        pos[K - 10:] = np.arange(10) * 1.0 + 0.5           # the last ten frames revisit frames 0..9
        far = np.abs(pos[:, None] - pos[None, :])
        d = np.minimum(d, 4.0 * far + rng.uniform(0, 3, (K, K)))
    d = np.round(d * 2) / 2                                 # half-pixel grid -> many exact ties
    d[rng.uniform(size=(K, K)) < 0.03] = np.inf
    d[rng.uniform(size=(K, K)) < 0.03] = 250.0
    d = np.tril(d, -1)
    return (d + d.T).astype(np.float32)


def topology_cases():
    """name -> (matrix kind, K, seed, max_factors, corr_impl, script).  Parameter sets follow the shipped configs
    (frontend: window 25, radius 2, nms 1, thresh 16, max_factors 75 / backend: radius 1, nms 5, thresh 25 /
    loop: window 25, radius 3, nms 12, thresh 25; /root/reference/configs/mono_point_slam.yaml)"""
    cases = {}
    # bootstrap + sliding window of the frontend, with ageing, the factor limit, inactive edges, bad edges and a
    # culled keyframe
    K = 14
    s = [["counter", 8], ["neigh", 0, 8, 3], ["age", 8], ["prox", 0, 0, 2, 2, 0.25, 16.0, False], ["age", 8],
         ["rm_lt", 4, True]]
    for t in range(9, K + 1):
        s += [["counter", t], ["rm_age", 20, True], ["prox", t - 5, max(t - 25, 0), 2, 1, 0.75, 16.0, True], ["age", 8]]
        if t == 10:
            s += [["weights"], ["filter_edges"]]
        if t == 11:
            s += [["rm_keyframe", t - 1]]
        s += [["age", 4]]
    cases["frontend_window"] = ("smooth", K, 11, 24, "volume", s)
    # small factor budget: every add runs into `max_factors` (oldest edges move to the inactive list)
    K = 12
    s = [["counter", 6], ["neigh", 0, 6, 3], ["age", 3]]
    for t in range(7, K + 1):
        s += [["counter", t], ["prox", max(t - 5, 0), 0, 1, 2, 0.5, 30.0, True], ["age", 2],
              ["add", [t - 1, 0], [0, t - 1], True], ["age", 1]]
    cases["factor_limit"] = ("smooth", K, 12, 16, "volume", s)
    # hand-made duplicate / self / repeated edges through add_factors, then removal of keyframes at either end
    s = [["counter", 7], ["add", [0, 1, 1, 2, 0], [1, 0, 0, 3, 1], False], ["add", [2, 3, 3], [3, 2, 3], False],
         ["age", 5], ["rm_age", 4, True], ["add", [0, 4, 5], [1, 5, 4], False], ["rm_keyframe", 0], ["rm_keyframe", 3]]
    cases["duplicates"] = ("smooth", 7, 13, -1, "alt", s)
    # global BA edge selection (dense_ba), several parameter sets on one matrix
    K = 36
    s = [["backprox", 0, K, 5, 1, 25.0, 6 * K, 0.75, None, False]]
    cases["backend_dense"] = ("loop", K, 14, 6 * K, "alt", s)
    s = [["backprox", 0, K, 2, 2, 40.0, 8 * K, 0.75, None, False]]
    cases["backend_dense_wide"] = ("loop", K, 15, 8 * K, "alt", s)
    s = [["backprox", 0, K, 3, 1, 25.0, 40, 0.75, None, False]]
    cases["backend_dense_budget"] = ("loop", K, 16, 40, "alt", s)
    s = [["backprox", 0, 2, 1, 1, 0.1, 100, 0.75, None, False]]            # fewer than 3 edges -> returns 0
    cases["backend_too_few"] = ("smooth", 8, 17, 100, "alt", s)
    # loop closure (loop_ba): local edges first, then loop candidates into the last 25 frames
    s = [["neigh", K - 6, K, 2], ["backprox", 0, K, 12, 1, 25.0, 200 - 18, 0.75, K - 25, True]]
    cases["loop_closure"] = ("loop", K, 18, 200, "alt", s)
    s = [["neigh", K - 6, K, 2], ["backprox", 0, K, 2, 1, 30.0, 200 - 18, 0.75, K - 25, True]]
    cases["loop_closure_nms2"] = ("loop", K, 19, 200, "alt", s)
    s = [["backprox", 0, 20, 12, 1, 25.0, 200, 0.75, 5, True]]              # no pair more than 20 apart -> 0 loop edges
    cases["loop_none"] = ("smooth", 20, 20, 200, "alt", s)
    return cases


# ------------------------------------------------------------------------------------------------------------------
# recording mocks for the DRIVER schedules (Frontend, Backend, PoseTrajectoryFiller)
# ------------------------------------------------------------------------------------------------------------------
def _plain(v):
    if isinstance(v, torch.Tensor):
        if v.dtype == torch.bool:
            return [bool(x) for x in v.reshape(-1).tolist()]
        if v.numel() == 1 and v.dim() == 0:
            return v.item()
        return v.reshape(-1).tolist()
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


class Trace:
    def __init__(self, plan):
        self.events = []
        self.plan = {k: list(v) for k, v in plan.items()}
        self.graphs = 0

    def add(self, who, what, **args):
        self.events.append([who, what, {k: _plain(v) for k, v in args.items()}])

    def take(self, key, default):
        q = self.plan.get(key)
        return q.pop(0) if q else default


class RecordingGraph:
    """stands in for FactorGraph under the drivers: records every call with its arguments bound to the reference's
    parameter names and keeps just enough edge state (ii / jj / age / corr) for the drivers' own decisions"""
    trace = None                                            # set by the harness before the driver is constructed

    def __init__(self, video, update_op, device="cuda:0", corr_impl="volume", max_factors=-1, **extra):
        tr = RecordingGraph.trace
        self.name = f"g{tr.graphs}"
        tr.graphs += 1
        tr.add(self.name, "__init__", corr_impl=corr_impl, max_factors=max_factors)
        self.video, self.update_op, self.device, self.corr_impl, self.max_factors = video, update_op, device, corr_impl, max_factors
        long0 = lambda: torch.zeros(0, dtype=torch.long)
        self.ii, self.jj, self.age = long0(), long0(), long0()
        self.corr = self.net = self.inp = None
        self.target = torch.zeros(1, 0, 1, 1, 2)
        self.weight = torch.zeros(1, 0, 1, 1, 2)
        self._topo = 0

    def _rec(self, what, **args):
        RecordingGraph.trace.add(self.name, what, **args)

    def _append(self, ii, jj):
        have = set(zip(self.ii.tolist(), self.jj.tolist()))
        new = [(i, j) for i, j in zip(ii, jj) if (i, j) not in have]
        if new:
            ni, nj = torch.tensor(new, dtype=torch.long).unbind(-1)
            self.ii = torch.cat([self.ii, ni])
            self.jj = torch.cat([self.jj, nj])
            self.age = torch.cat([self.age, torch.zeros_like(ni)])
            self.corr = "volume"
            self.net = torch.zeros(1, len(self.ii), 1, 1, 1)
            self.target = torch.zeros(1, len(self.ii), 1, 1, 2)
            self.weight = torch.zeros(1, len(self.ii), 1, 1, 2)

    def add_factors(self, ii, jj, remove=False):
        self._rec("add_factors", ii=ii, jj=jj, remove=remove)
        self._append(_plain(ii), _plain(jj))

    def add_neighborhood_factors(self, t0, t1, r=3):
        self._rec("add_neighborhood_factors", t0=t0, t1=t1, r=r)
        e = [(i, j) for i in range(t0, t1) for j in range(t0, t1) if 0 < abs(i - j) <= r]
        self._append([a for a, _ in e], [b for _, b in e])

    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False):
        self._rec("add_proximity_factors", t0=t0, t1=t1, rad=rad, nms=nms, beta=beta, thresh=thresh, remove=remove)
        t = self.video.counter.value
        e = [(i, j) for i in range(max(t0, 0), t) for j in range(max(i - rad - 1, 0), i)]
        self._append([a for a, _ in e] + [b for _, b in e], [b for _, b in e] + [a for a, _ in e])

    def add_backend_proximity_factors(self, t_start, t_end, nms, radius, thresh, max_factors, beta, t_start_loop=None,
                                      loop=False):
        self._rec("add_backend_proximity_factors", t_start=t_start, t_end=t_end, nms=nms, radius=radius, thresh=thresh,
                  max_factors=max_factors, beta=beta, t_start_loop=t_start_loop, loop=loop, edges_before=len(self.ii))
        return RecordingGraph.trace.take("edge_num", 0)

    def rm_factors(self, mask, store=False):
        self._rec("rm_factors", mask=mask, store=store)
        keep = ~mask.cpu()
        self.ii, self.jj, self.age = self.ii[keep], self.jj[keep], self.age[keep]

    def rm_keyframe(self, ix):
        self._rec("rm_keyframe", ix=ix)
        m = (self.ii == ix) | (self.jj == ix)
        self.ii[self.ii >= ix] -= 1
        self.jj[self.jj >= ix] -= 1
        self.ii, self.jj, self.age = self.ii[~m], self.jj[~m], self.age[~m]

    def update(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False, opt_type="pose_depth"):
        self._rec("update", t0=t0, t1=t1, itrs=itrs, use_inactive=use_inactive, EP=EP, motion_only=motion_only,
                  opt_type=opt_type)
        self.age += 1

    def update_lowmem(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, steps=8, enable_wq=True):
        self._rec("update_lowmem", t0=t0, t1=t1, itrs=itrs, use_inactive=use_inactive, EP=EP, steps=steps,
                  enable_wq=enable_wq, edges=len(self.ii))

    def clear_edges(self):
        self._rec("clear_edges")
        self.ii = self.jj = self.age = None


class DriverVideo:
    """the DepthVideo surface Frontend / Backend / PoseTrajectoryFiller touch; `distance` answers from the plan"""

    def __init__(self, trace, buffer=64, ht=2, wd=3, device="cpu"):
        self.trace = trace
        self.device = device
        self.counter = Counter(0)
        self.ht, self.wd, self.down_scale = ht * 8, wd * 8, 8
        k = torch.arange(buffer, dtype=torch.float32)
        self.poses = torch.zeros(buffer, 7)
        self.poses[:, 6] = 1.0
        self.poses[:, 0] = 0.01 * k
        self.disps = (1.0 + 0.125 * k)[:, None, None] * torch.linspace(0.5, 1.5, ht * wd).reshape(1, ht, wd)
        self.timestamp = 4.0 * k
        self.intrinsics = torch.ones(buffer, 4)
        self.fmaps = torch.zeros(buffer, 1, 4, ht, wd)
        self.stored = []

    @contextlib.contextmanager
    def get_lock(self):
        yield

    def distance(self, ii=None, jj=None, beta=0.3, bidirectional=True):
        self.trace.add("video", "distance", ii=ii, jj=jj, beta=beta, bidirectional=bidirectional)
        return torch.tensor([self.trace.take("distance", 10.0)])

    def set_dirty(self, index_start, index_end):
        self.trace.add("video", "set_dirty", index_start=int(index_start), index_end=int(index_end))

    def update_valid_depth_mask(self, *a, **k):
        self.trace.add("video", "update_valid_depth_mask")

    def normalize(self):
        self.trace.add("video", "normalize")

    def __setitem__(self, index, item):
        # (tstamp, image, pose, disp, depth, intrinsics, fmap[, net, inp]) like depth_video.py:61-98
        self.trace.add("video", "__setitem__", start=index.start, stop=index.stop, timestamps=item[0],
                       disp=item[3], has_depth=item[4] is not None, intrinsics_row0=item[5][0])
        n = index.stop - index.start
        self.timestamp[index] = torch.as_tensor(item[0]).float()
        if item[2] is not None:
            self.poses[index] = item[2].reshape(n, 7)


def driver_cfg(enable_loop, normalize=False, device="cpu"):
    return {"device": device, "setting": "s", "scene": "x", "data": {"output": "/tmp"},
            "tracking": {"max_age": 20, "warmup": 8, "beta": 0.75,
                         "frontend": {"nms": 1, "keyframe_thresh": 4.0, "window": 10, "thresh": 16.0, "radius": 2,
                                      "max_factors": 75, "enable_loop": enable_loop},
                         "backend": {"thresh": 25.0, "radius": 1, "nms": 5, "normalize": normalize, "loop_window": 25,
                                     "loop_thresh": 25.0, "loop_radius": 3, "loop_nms": 12}}}


class DummyNet:
    update = "update_op"
    cnet = "cnet"

    @staticmethod
    def fnet(image):
        b, n = image.shape[:2]
        return torch.zeros(b, n, 4, image.shape[-2] // 8, image.shape[-1] // 8)


def frontend_scenarios():
    """name -> (enable_loop, counter values fed call by call, plan).  The plan scripts what the mocks answer:
    `distance` = the redundancy test of each update (frontend.py:56-59, keyframe_thresh 4.0), `edge_num` = what
    add_backend_proximity_factors reports to loop_ba."""
    return {
        # below warm-up (no-op), bootstrap, kept keyframe, culled keyframe, idle call, kept keyframe
        "no_loop": (False, [3, 8, 9, 10, 9, 10], {"distance": [9.0, 1.5, 7.0]}),
        # loop closure enabled: window 10 -> first update with 11 keyframes tries loop_ba; first attempt finds no
        # edges (falls back to four local iterations), second closes a loop, third frame is culled
        "loop": (True, [8, 9, 10, 11, 12, 13], {"distance": [9.0, 9.0, 9.0, 9.0, 2.0], "edge_num": [0, 57]}),
    }


def run_frontend(frontend_cls, graph_patch, scenario, make_video=DriverVideo):
    """graph_patch(cls): installs RecordingGraph where the driver under test looks FactorGraph up"""
    enable_loop, counters, plan = scenario
    tr = Trace(plan)
    RecordingGraph.trace = tr
    graph_patch(RecordingGraph)
    video = make_video(tr)
    fe = frontend_cls(DummyNet(), video, driver_cfg(enable_loop))
    for c in counters:
        video.counter.value = c
        tr.add("harness", "call", counter=c)
        fe()
        t1 = fe.t1
        tr.add("harness", "after", t1=t1, counter=video.counter.value, initialized=bool(fe.is_initialized),
               pose_t1=[round(float(v), 6) for v in video.poses[t1].tolist()],
               disp_t1=round(float(video.disps[t1].mean()), 6))
    return tr.events


def backend_scenarios():
    """name -> (normalize, counter, call, kwargs, plan)"""
    return {
        "dense_ba": (False, 20, "dense_ba", {"steps": 7}, {"edge_num": [140]}),
        "dense_ba_normalize_no_wq": (True, 12, "dense_ba", {"steps": 2, "enable_wq": False}, {"edge_num": [66]}),
        "dense_ba_no_edges": (False, 5, "dense_ba", {}, {"edge_num": [0]}),
        "loop_ba_fresh": (False, 40, "loop_ba", {"t_start": 0, "t_end": 40, "steps": 4}, {"edge_num": [33]}),
        "loop_ba_local_graph": (False, 30, "loop_ba", {"t_start": 0, "t_end": 30, "steps": 4, "motion_only": False,
                                                       "local_graph": "neigh", "enable_wq": True}, {"edge_num": [12]}),
        "loop_ba_short": (False, 6, "loop_ba", {"t_start": 0, "t_end": 6, "steps": 2, "local_graph": "neigh"},
                          {"edge_num": [0]}),
    }


def run_backend(backend_cls, graph_patch, scenario, make_video=DriverVideo):
    normalize, counter, call, kwargs, plan = scenario
    tr = Trace(plan)
    RecordingGraph.trace = tr
    graph_patch(RecordingGraph)
    video = make_video(tr)
    video.counter.value = counter
    be = backend_cls(DummyNet(), video, driver_cfg(False, normalize))
    kwargs = dict(kwargs)
    if kwargs.get("local_graph") == "neigh":
        lg = RecordingGraph(video, "update_op")
        lg.add_neighborhood_factors(max(counter - 6, 0), counter, r=2)
        lg.age += 3
        kwargs["local_graph"] = lg
    ret = getattr(be, call)(**kwargs)
    tr.add("harness", "return", value=[int(v) for v in ret])
    return tr.events


class Stream:
    """an image stream like datasets.BaseDataset: items (timestamp, image [1,3,H,W], depth, intrinsic)"""

    def __init__(self, stamps, H=16, W=24):
        self.stamps = list(stamps)
        self.H, self.W = H, W

    def get_intrinsic(self):
        return torch.tensor([20.0, 20.0, 12.0, 8.0])

    def __len__(self):
        return len(self.stamps)

    def __iter__(self):
        for t in self.stamps:
            yield (t, torch.full((1, 3, self.H, self.W), 0.5), None, None)


def filler_scenarios():
    """name -> (number of keyframes, frame time stamps).  Keyframe k carries time stamp 4 k (DriverVideo)"""
    return {"one_batch": (6, [1, 2, 3, 5, 9, 10, 19, 20, 21]),
            "two_batches_and_rest": (9, list(range(0, 35)))}


def run_filler(filler_cls, graph_patch, scenario, make_video=DriverVideo):
    n_kf, stamps = scenario
    tr = Trace({})
    RecordingGraph.trace = tr
    graph_patch(RecordingGraph)
    video = make_video(tr)
    video.counter.value = n_kf

    class Printer:
        def print(self, *a, **k):
            pass

    filler = filler_cls(net=DummyNet(), video=video, printer=Printer(), device="cpu")
    out = filler(Stream(stamps))
    tr.add("harness", "return", poses=int(out.data.shape[0]), counter=video.counter.value)
    return tr.events


def normalise_events(events):
    """JSON round trip (tuples -> lists, ints stay ints): both sides are compared in this form"""
    import json
    return json.loads(json.dumps(events))


# ------------------------------------------------------------------------------------------------------------------
# call signatures
# ------------------------------------------------------------------------------------------------------------------
def signature_of(fn):
    """[[name, kind, default-or-None-marker], ...] without `self`; defaults as repr strings"""
    out = []
    for p in inspect.signature(fn).parameters.values():
        if p.name == "self":
            continue
        out.append([p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)])
    return out
