"""Tracking + mapping composed on one stream of frames (BASELINE config 3: a full sequence on one GPU).

The reference runs a tracker process and a mapper process around shared memory (src/slam.py:119-126 spawns them;
src/tracker.py:33-77 is the per-frame loop `MotionFilter.track -> Frontend() -> periodic Backend.dense_ba`;
src/mapper.py:517-684 is the per-keyframe loop `add_neural_points -> optimise features + decoders against the keyframe's
depth and colour`; slam.py / tracker.py end with a final global BA).  Process orchestration, the mono-depth network, pose
evaluation and the mapper's keyframe-selection heuristics are out of scope; what is in scope are the
callers' CALL SHAPES, because they decide whether the hot-path pieces compose: the frontend's local graph with its
hipGraph replays and slot arena, the backend's throw-away low-memory graphs over the same video buffers, the cloud's cell
list being rebuilt under the renderer, the training path feeding FeatureAdam with tables that grow between keyframes.
`SequenceRunner` is that composition in one process, in the order the reference's two loops interleave when the mapper
keeps up with the tracker (every kept keyframe is mapped before the next frame is tracked).
"""
import time

import torch

from .backend import Backend
from .common import get_rays_from_uv
from .frontend import Frontend
from .motion_filter import MotionFilter
from .neural_point import se3_inv
from .render_train import FeatureAdam


def pose_matrix(pose7):
    """[tx ty tz qx qy qz qw] -> 4x4"""
    t, q = pose7[:3], pose7[3:]
    x, y, z, w = q.unbind(-1)
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)]),
                     torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)]),
                     torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)])])
    M = torch.eye(4, device=pose7.device, dtype=pose7.dtype)
    M[:3, :3] = R
    M[:3, 3] = t
    return M


class SequenceRunner:
    """net: DroidNet-like (fnet, cnet, update); video: DepthVideo; npc / decoders / renderer: the neural point cloud side.
    mono_depth_fn(tstamp, image) -> [H,W] depth prior (the reference reads it from the estimator's output directory)."""

    def __init__(self, net, video, cfg, npc, decoders, renderer, mono_depth_fn, use_graphs=True, ba_every=4, ba_steps=2,
                 map_iters=20, map_rays=1000, add_stride=8, seed=43):
        self.net, self.video, self.cfg = net, video, cfg
        self.npc, self.decoders, self.renderer = npc, decoders, renderer
        dev = cfg["device"]
        self.device = dev
        self.filter = MotionFilter(net, video, cfg, thresh=cfg["tracking"].get("motion_filter", {}).get("thresh", 0.0),
                                   device=dev, mono_depth_fn=mono_depth_fn)
        self.frontend = Frontend(net, video, cfg, use_graphs=use_graphs)
        self.backend = Backend(net, video, cfg)
        self.ba_every, self.ba_steps = ba_every, ba_steps
        self.map_iters, self.map_rays, self.add_stride = map_iters, map_rays, add_stride
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self.images = {}                      # keyframe index -> [3,H,W] colour image (what video.images stores in the reference)
        self.mapped = 0                       # keyframes [0, mapped) are in the cloud
        self.last_ba = 0
        self.losses = []                      # per mapped keyframe: (first, last) loss of its mapping iterations
        self.timing = {"track_ms": [], "map_iter_ms": [], "ba_ms": [], "kept": []}
        # record the mapping iteration into a hipGraph per keyframe (map_keyframe).  Off by default: built and measured in
        # round 5 (tools/prof_map_graph.py) - recording 3.3 ms per keyframe, a replay 2.26 ms against 2.0-2.5 ms for an eager
        # iteration at 1000 rays: the iteration is ~110 small launches whose device-side latency a replay does not shorten
        # (the sum of its kernel durations IS its wall time, profiles/r04_train_kernel_stats.csv), not host-bound as the
        # round-4 notes had it.  What would shorten it is fewer launches (fused layers in csrc/train.hip).
        self.map_graph = False
        self.map_graph_stats = {"captures": 0, "replays": 0}
        self.init_state = None                # callable(k, tstamp): the tracker's initial guess for a new keyframe (tests)

    # ---- tracker.py:33-77 ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def track(self, tstamp, image, intrinsics):
        """one frame: motion filter, local BA-update iterations, periodic global BA.  Returns True if the frame was kept"""
        t0 = time.perf_counter()
        n_before = self.video.counter.value
        self.filter.track(tstamp, image, intrinsics)
        appended = self.video.counter.value > n_before
        if appended:
            self.images[n_before] = image[0].to(self.device)
            if self.init_state is not None:
                self.init_state(n_before, tstamp)
        self.frontend()
        kept = appended and self.video.counter.value > n_before
        cur = self.video.counter.value
        torch.cuda.synchronize()
        self.timing["track_ms"].append(1e3 * (time.perf_counter() - t0))      # motion filter + frontend of this frame
        self.timing["kept"].append(bool(kept))
        if self.frontend.is_initialized and cur - self.last_ba >= self.ba_every and self.ba_every > 0:
            tb = time.perf_counter()
            self.backend.dense_ba(self.ba_steps)
            torch.cuda.synchronize()
            self.timing["ba_ms"].append(1e3 * (time.perf_counter() - tb))
            self.last_ba = cur
        return kept

    # ---- mapper.py:517-684 (one keyframe) ---------------------------------------------------------------------------
    def _keyframe_view(self, k):
        """depth map and camera-to-world matrix of keyframe k (constant while the keyframe is mapped)"""
        v = self.video
        depth = torch.where(v.disps_up[k] > 0, 1.0 / v.disps_up[k].clamp_min(1e-6), torch.zeros_like(v.disps_up[k]))
        c2w = pose_matrix(se3_inv(v.poses[k]))
        # the neural point cloud lives in the OpenGL camera convention of the renderer (common.py:302-322)
        c2w = c2w.clone()
        c2w[:3, 1] *= -1
        c2w[:3, 2] *= -1
        return depth, c2w

    def _keyframe_rays(self, k, stride=1, count=None, view=None, pix=None):
        """pixels of keyframe k with a valid depth -> rays, depth, colour, pixel ids (common.py:39-54 ray convention).
        view: `_keyframe_view(k)` computed by the caller; pix: the (ii, jj) pixel draw of this call, already on the device
        (the mapping loop draws all its iterations at once: a per-iteration host-to-device copy from pageable memory blocks
        the host until the stream has drained)"""
        v = self.video
        H, W = v.ht, v.wd
        cam = self.renderer
        depth, c2w = view if view is not None else self._keyframe_view(k)
        if pix is not None:
            ii, jj = pix
        elif count is None:
            jj, ii = torch.meshgrid(torch.arange(stride // 2, H, stride, device=self.device),
                                    torch.arange(stride // 2, W, stride, device=self.device), indexing="ij")
            ii, jj = ii.reshape(-1), jj.reshape(-1)
        else:
            ii = torch.randint(0, W, (count,), generator=self.gen).to(self.device)
            jj = torch.randint(0, H, (count,), generator=self.gen).to(self.device)
        ro, rd = get_rays_from_uv(ii.float(), jj.float(), c2w, cam.fx, cam.fy, cam.cx, cam.cy, self.device)
        d = depth[jj, ii]
        col = self.images[k][:, jj, ii].t().contiguous()
        radius = None
        if self.npc.use_dynamic_radius:
            radius = torch.full_like(d, 0.5 * (self.npc.radius_add + self.npc.radius_query))
        return ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous(), d, col, ii, jj, radius

    def map_keyframe(self, k):
        """seed the keyframe's points, then `map_iters` iterations on `map_rays` of its pixels (depth + colour L1,
        mapper.py:497-505; learning rates of the colour stage, mapper.py:412-414)"""
        npc, dec, ren = self.npc, self.decoders, self.renderer
        with torch.no_grad():
            ro, rd, d, col, ii, jj, radius = self._keyframe_rays(k, stride=self.add_stride)
            npc.add_neural_points(ro, rd, d, col, k, ii, jj, dynamic_radius=radius)
        if npc.pts_num() == 0:
            return None
        geo = npc.geo_feats.detach().clone().requires_grad_(True)
        col_f = npc.col_feats.detach().clone().requires_grad_(True)
        dec.train()
        for p in dec.parameters():
            p.requires_grad_(True)
        with torch.no_grad():
            view = self._keyframe_view(k)
            # every iteration's pixel draw in one transfer ([iteration][ii | jj][ray], the generator's order per iteration)
            draws = torch.stack([torch.stack([torch.randint(0, self.video.wd, (self.map_rays,), generator=self.gen),
                                              torch.randint(0, self.video.ht, (self.map_rays,), generator=self.gen)])
                                 for _ in range(self.map_iters)]).to(self.device) if self.map_iters else None
            # one look at the keyframe's depth map (a host round trip per KEYFRAME): if every pixel carries a depth, no
            # iteration needs the host - the renderer skips its per-batch zero-depth check and the iteration can be recorded
            all_depth = bool((view[0] > 0).all())
        use_graph = bool(getattr(self, "map_graph", True)) and all_depth and self.map_iters >= 4 and \
            str(self.device).startswith("cuda") and getattr(ren, "use_train_path", True)
        opt = FeatureAdam([{"params": list(dec.parameters()), "lr": 0.005}, {"params": [geo], "lr": 0.005},
                           {"params": [col_f], "lr": 0.005}], capturable=use_graph)
        self.last_optimizer = opt                            # (tests: step bookkeeping)
        pix = draws[0].clone() if self.map_iters else None   # the iteration reads its pixels from this buffer

        def iteration():
            with torch.no_grad():
                ro, rd, d, gt_col, _, _, radius = self._keyframe_rays(k, view=view, pix=(pix[0], pix[1]))
            opt.zero_grad()
            depth, _, colour, _, counts = ren.render_batch_ray(npc, dec, rd, ro, self.device, "color", gt_depth=d,
                                                               npc_geo_feats=geo, npc_col_feats=col_f,
                                                               cloud_pos=npc.cloud_pos(), dynamic_r_query=radius)
            # (masked sums instead of `x[seen].sum()`: boolean indexing sizes its result on the host - one device round trip
            # in the forward and one in the backward pass of every iteration)
            seen = ((counts > 0) & (d > 0)).to(depth.dtype)
            loss = (torch.abs(d - depth) * seen).sum() + 0.5 * (torch.abs(gt_col - colour) * seen[:, None]).sum()
            loss = loss / seen.sum().clamp_min(1)
            loss.backward()
            opt.step()
            return loss.detach()

        first = last = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ren.assume_depth = all_depth
        try:
            graph = None
            for it in range(self.map_iters):
                if it:
                    pix.copy_(draws[it])
                if graph is not None:
                    graph.replay()
                    continue
                if use_graph and it == 1:
                    # Iteration 0 ran eagerly (it sizes the scratch arenas, creates the Adam moments and sends the pointer
                    # table); iteration 1 is RECORDED - forward, masked loss, backward (incl. its second stream for the weight
                    # gradients), Adam with the step count in device memory - and replayed for itself and every later one
                    # (mapper.py:586-624 runs 150-1500 of them per keyframe on the same tensors).
                    try:
                        graph, last = self._capture_iteration(iteration)
                    except Exception as exc:      # a failed recording must not take the mapper down
                        import warnings
                        warnings.warn(f"hipGraph capture of the mapping iteration failed ({exc!r}); running eagerly")
                        graph, use_graph = None, False
                        opt.capturable_fallback()
                    if graph is not None:
                        graph.replay()
                        continue
                last = iteration()
                if first is None:
                    first = last.clone()
            if graph is not None:
                # host bookkeeping: the eager iteration and the recording itself each counted one step, the device did
                # 1 + (map_iters - 1)
                opt.advance(self.map_iters - 2)
                last = last.clone()
                self.map_graph_stats["captures"] += 1
                self.map_graph_stats["replays"] += self.map_iters - 1
        finally:
            ren.assume_depth = False
        torch.cuda.synchronize()
        self.timing["map_iter_ms"].append(1e3 * (time.perf_counter() - t0) / max(self.map_iters, 1))
        with torch.no_grad():
            npc.update_geo_feats(geo.detach())
            npc.update_col_feats(col_f.detach())
        dec.eval()
        self.losses.append((float(first), float(last)))
        return self.losses[-1]

    def _capture_iteration(self, iteration):
        """record one mapping iteration into a hipGraph (one memory pool for all keyframes of this runner, kept open by a
        one-node keeper graph as in FactorGraph._capture); returns (graph, the iteration's loss tensor inside the pool)"""
        dev = torch.device(self.device)
        cap = getattr(self, "_map_capture_stream", None)
        if cap is None:
            cap = self._map_capture_stream = torch.cuda.Stream(dev)
            self._map_pool = torch.cuda.graph_pool_handle()
            self._map_keeper = torch.cuda.CUDAGraph()
            with torch.cuda.stream(cap):
                self._map_keeper.capture_begin(pool=self._map_pool)
                self._map_keeper_buf = torch.zeros(1, device=dev)
                self._map_keeper.capture_end()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            graph.capture_begin(pool=self._map_pool)
            try:
                loss = iteration()
            finally:
                graph.capture_end()
        torch.cuda.current_stream(dev).wait_stream(cap)
        return graph, loss

    def map_pending(self):
        """map every keyframe the frontend has finished with (its redundancy test culls a frame inside the same call that
        appended it, tracker.py:56-69: what is below the window's end afterwards is final)"""
        done = self.frontend.t1 if self.frontend.is_initialized else 0
        while self.mapped < done:
            self.map_keyframe(self.mapped)
            self.mapped += 1

    # ---- the stream -------------------------------------------------------------------------------------------------
    def run(self, frames, intrinsics, final_ba_steps=7):
        """frames: iterable of (tstamp, image [1,3,H,W] in [0,1]).  Returns a summary dict."""
        for tstamp, image in frames:
            self.track(tstamp, image, intrinsics)
            self.map_pending()
        with torch.no_grad():
            n, n_edges = self.backend.dense_ba(final_ba_steps)             # the final global BA of slam.py / tracker.py
        self.map_pending()
        torch.cuda.synchronize()
        return {"keyframes": int(self.video.counter.value), "final_ba_edges": int(n_edges), "mapped": self.mapped,
                "points": int(self.npc.pts_num()), "losses": list(self.losses), "timing": self.timing}


# ---- the synthetic 640x480 stream of the config-3 test and of bench.py's `sequence` entry --------------------------------
def synthetic_cfg(device, buffer, H=480, W=640):
    return {
        "cam": {"H_out": H, "W_out": W}, "device": str(device), "setting": "t", "scene": "seq", "data": {"output": "/tmp"},
        "mono_prior": {"predict_online": False}, "mapping": {"every_frame": 5},
        "tracking": {
            "buffer": buffer, "beta": 0.75, "warmup": 8, "max_age": 50, "mono_thres": 0.1,
            "multiview_filter": {"thresh": 0.25, "visible_num": 2}, "store_images": False,
            "motion_filter": {"thresh": -1.0},     # every frame becomes a keyframe (a zeroed flow head predicts no motion)
            "frontend": {"enable_loop": False, "keyframe_thresh": 0.0, "thresh": 16.0, "window": 25, "radius": 1,
                         "nms": 1, "max_factors": 75},
            "backend": {"BA_type": "DSPO", "thresh": 25.0, "radius": 1, "nms": 5, "normalize": False,
                        "loop_window": 25, "loop_thresh": 25.0, "loop_radius": 1, "loop_nms": 12}},
        "pointcloud": {"nn_weighting": "distance", "use_dynamic_radius": True, "min_nn_num": 2, "nn_num": 8,
                       "radius_query": 0.08, "radius_add": 0.04, "radius_min": 0.02},
        "rendering": {"N_surface": 10, "near_end_surface": 0.95, "far_end_surface": 1.05, "sample_near_pcl": True,
                      "sigmoid_coef": 0.1, "near_end": 0.3},
        "model": {"encode_rel_pos_in_col": True, "encode_viewd": True, "c_dim": 32}}


def synthetic_images(K, H=480, W=640, seed=5):
    """smooth random textures in [0, 1] (low-resolution noise, bilinearly upsampled): [K,3,H,W]"""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(K, 3, H // 16, W // 16, generator=g)
    return torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False).clamp(0, 1)


def synthetic_runner(device, K, zero_flow_head=True, map_iters=20, map_rays=1000, use_graphs=True, H=480, W=640,
                     buffer=None):
    """SequenceRunner over the keyframe arc of synth.keyframe_graph (the bench's graph G8 continued to K frames): seed-43
    default-init DroidNet and decoders, the mono prior = the true depth under an affine distortion, the tracker's initial
    guess of a new keyframe = its generating pose and disparity.  zero_flow_head: the last layer of the flow head is
    zeroed, which makes the generating trajectory a fixed point of the tracking loop (there are no trained weights
    here).  -> (runner, dict(poses, disps, cfg, video, npc, intrinsics))"""
    import types

    import numpy as np

    from . import synth
    from .decoder import POINT
    from .depth_video import DepthVideo
    from .droid_net import DroidNet
    from .neural_point import NeuralPointCloud
    from .renderer import Renderer
    h, w = H // 8, W // 8
    cfg = synthetic_cfg(device, max(K + 2, buffer or 0), H, W)    # buffer: tracking.buffer of the shipped configs is 400-600
    g = synth.keyframe_graph(K=K, h=h, w=w, radius=3)
    torch.manual_seed(43)
    net = DroidNet().to(device).eval()
    if zero_flow_head:
        with torch.no_grad():
            net.update.delta[2].weight.zero_()
            net.update.delta[2].bias.zero_()
    video = DepthVideo(cfg)
    npc = NeuralPointCloud(cfg)
    torch.manual_seed(43)
    dec = POINT(cfg, use_view_direction=True).eval().to(device)
    sx, sy = W / 640.0, H / 480.0
    cam = types.SimpleNamespace(H=H, W=W, fx=320.0 * sx, fy=320.0 * sy, cx=319.5 * sx, cy=239.5 * sy)
    ren = Renderer(cfg, cam)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    disps = t(g["disps"])                                        # [K,h,w] generating disparities
    full = torch.nn.functional.interpolate(1.0 / disps[:, None], size=(H, W), mode="bilinear", align_corners=False)[:, 0]

    def mono(tstamp, image):
        return (full[int(tstamp)] * 1.25 + 0.1).to(device)

    run = SequenceRunner(net, video, cfg, npc, dec, ren, mono, use_graphs=use_graphs, ba_every=4, ba_steps=2,
                         map_iters=map_iters, map_rays=map_rays, add_stride=8)
    poses = t(g["poses"])

    def init_state(k, tstamp=None):
        video.poses[k] = poses[k]
        video.disps[k] = disps[k]
        video.disps_up[k] = 1.0 / full[k]
    run.init_state = init_state
    intr = torch.tensor([cam.fx, cam.fy, cam.cx, cam.cy])
    return run, dict(poses=poses, disps=disps, cfg=cfg, video=video, npc=npc, intrinsics=intr)


def synthetic_long_runner(device, n_frames=220, leg=60, repeat_every=9, map_iters=20, map_rays=1000, buffer=512, ba_every=20,
                          H=480, W=640):
    """Config 3 beyond the 13-frame miniature: a stream of `n_frames` 640x480 frames whose camera walks the arc of
    synth.keyframe_graph forth and back (a triangle wave of period 2 * leg: frame 2 * leg stands where frame 0 stood, so the
    loop-closure search finds pairs more than 20 keyframes apart), with every `repeat_every`-th frame a repeat of its
    predecessor (zero induced flow: the frontend's redundancy test culls it, rm_keyframe shifts the buffers), loop closure
    enabled, a global BA every `ba_every` keyframes, `map_iters` mapping iterations per kept keyframe, a 512-frame buffer.
    Flow head zeroed: the generating trajectory is a fixed point.  -> (runner, dict(...), frames iterator factory)"""
    import types

    import numpy as np

    from . import synth
    from .decoder import POINT
    from .depth_video import DepthVideo
    from .droid_net import DroidNet
    from .neural_point import NeuralPointCloud
    from .renderer import Renderer
    h, w = H // 8, W // 8
    cfg = synthetic_cfg(device, buffer, H, W)
    cfg["tracking"]["frontend"].update({"enable_loop": True, "keyframe_thresh": 0.5})
    g = synth.keyframe_graph(K=leg + 1, h=h, w=w, radius=3)
    # arc position of stream frame f: triangle wave; every repeat_every-th frame repeats its predecessor
    pos, u = [], 0
    for f in range(n_frames):
        if f and f % repeat_every == 0:
            pos.append(pos[-1])
            continue
        k = u % (2 * leg)
        pos.append(k if k <= leg else 2 * leg - k)
        u += 1
    torch.manual_seed(43)
    net = DroidNet().to(device).eval()
    with torch.no_grad():
        net.update.delta[2].weight.zero_()
        net.update.delta[2].bias.zero_()
    video = DepthVideo(cfg)
    npc = NeuralPointCloud(cfg)
    torch.manual_seed(43)
    dec = POINT(cfg, use_view_direction=True).eval().to(device)
    cam = types.SimpleNamespace(H=H, W=W, fx=320.0 * W / 640.0, fy=320.0 * H / 480.0, cx=319.5 * W / 640.0, cy=239.5 * H / 480.0)
    ren = Renderer(cfg, cam)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    disps, poses = t(g["disps"]), t(g["poses"])
    full = torch.nn.functional.interpolate(1.0 / disps[:, None], size=(H, W), mode="bilinear", align_corners=False)[:, 0]
    run = SequenceRunner(net, video, cfg, npc, dec, ren, lambda ts, im: (full[pos[int(ts)]] * 1.25 + 0.1), use_graphs=True,
                         ba_every=ba_every, ba_steps=2, map_iters=map_iters, map_rays=map_rays, add_stride=8)

    def init_state(k, tstamp):
        a = pos[int(tstamp)]
        video.poses[k] = poses[a]
        video.disps[k] = disps[a]
        video.disps_up[k] = 1.0 / full[a]
    run.init_state = init_state
    bank = synthetic_images(leg + 1, H, W)                                  # one texture per arc position

    def frames():
        for f in range(n_frames):
            yield f, bank[pos[f]:pos[f] + 1]
    intr = torch.tensor([cam.fx, cam.fy, cam.cx, cam.cy])
    return run, dict(poses=poses, disps=disps, cfg=cfg, video=video, npc=npc, intrinsics=intr, pos=pos), frames
