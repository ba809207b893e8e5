"""Local-window tracking driver (scope row N3) -- same interface as /root/reference/src/frontend.py:
`Frontend(net, video, cfg)()` is called once per incoming keyframe candidate.

It owns the sliding-window FactorGraph (correlation volumes, `corr_impl='volume'`) and issues exactly the
sequence of BA-update iterations the headline metric counts: bootstrap = 8 + 8 pose_depth iterations
(frontend.py:86-98), every later keyframe = 8 iterations alternating pose_depth / depth_scale, a redundancy
test on the newest frame pair, then 4 more iterations or a loop-closure BA (frontend.py:40-83).
"""
import torch

from .backend import Backend as LoopClosing
from .factor_graph import FactorGraph

_STAGES = ("pose_depth", "depth_scale")


class Frontend:
    def __init__(self, net, video, cfg, use_graphs=False):
        trk = cfg['tracking']
        fe = trk['frontend']
        self.video = video
        self.update_op = net.update
        self.t1 = 0                               # end of the local window
        self.is_initialized = False
        self.max_age = trk['max_age']
        self.iters1, self.iters2 = 8, 4
        self.warmup, self.beta = trk['warmup'], trk['beta']
        self.frontend_nms, self.keyframe_thresh = fe['nms'], fe['keyframe_thresh']
        self.frontend_window, self.frontend_thresh = fe['window'], fe['thresh']
        self.frontend_radius, self.frontend_max_factors = fe['radius'], fe['max_factors']
        self.enable_loop = fe['enable_loop']
        self.loop_closing = LoopClosing(net, video, cfg)
        # capture_after=6: a keyframe changes the edge set and each of the two alternating calls is then seen ~6 times - not
        # enough replays to pay for a capture (FactorGraph.update); a graph that survives longer (no new keyframe: a camera
        # standing still) is still recorded
        self.graph = FactorGraph(video, net.update, device=cfg['device'], corr_impl='volume',
                                 max_factors=self.frontend_max_factors, use_graphs=use_graphs, capture_after=6)

    def _refine(self, iterations):
        """DSPO: even iterations optimise poses + disparities, odd ones disparities + scale/shift"""
        for it in range(iterations):
            self.graph.update(None, None, use_inactive=True, opt_type=_STAGES[it % 2])

    def _update(self):
        video, graph = self.video, self.graph
        self.t1 += 1
        if graph.corr is not None:
            graph.rm_factors(graph.age > self.max_age, store=True)
        graph.add_proximity_factors(self.t1 - 5, max(self.t1 - self.frontend_window, 0), rad=self.frontend_radius,
                                    nms=self.frontend_nms, thresh=self.frontend_thresh, beta=self.beta,
                                    remove=True)
        self._refine(self.iters1)
        # is the newest keyframe redundant?  (mean flow to its predecessor below the threshold)
        flow = video.distance([self.t1 - 2], [self.t1 - 1], beta=self.beta, bidirectional=True)
        if flow.item() < self.keyframe_thresh:
            graph.rm_keyframe(self.t1 - 1)
            with video.get_lock():
                video.counter.value -= 1
                self.t1 -= 1
        else:
            cur_t = video.counter.value
            closed = 0
            if self.enable_loop and cur_t > self.frontend_window:
                _, closed = self.loop_closing.loop_ba(t_start=0, t_end=cur_t, steps=self.iters2, motion_only=False,
                                                      local_graph=graph, enable_wq=True)
                self.last_loop_t = cur_t
            if closed == 0:
                self._refine(self.iters2)
        # initial guess for the next frame: pose of the last keyframe, its mean disparity
        video.poses[self.t1] = video.poses[self.t1 - 1]
        video.disps[self.t1] = video.disps[self.t1 - 1].mean()
        video.set_dirty(int(graph.ii.min()), self.t1)

    def _initialize(self):
        video, graph = self.video, self.graph
        self.t1 = video.counter.value
        graph.add_neighborhood_factors(0, self.t1, r=3)
        for _ in range(8):
            graph.update(1, use_inactive=True, opt_type="pose_depth")
        graph.add_proximity_factors(0, 0, rad=2, nms=2, thresh=self.frontend_thresh, remove=False)
        for _ in range(8):
            graph.update(1, use_inactive=True, opt_type="pose_depth")
        video.poses[self.t1] = video.poses[self.t1 - 1].clone()
        video.disps[self.t1] = video.disps[self.t1 - 4:self.t1].mean()
        self.is_initialized = True
        self.last_pose = video.poses[self.t1 - 1].clone()
        self.last_disp = video.disps[self.t1 - 1].clone()
        self.last_time = video.timestamp[self.t1 - 1].clone()
        with video.get_lock():
            video.set_dirty(0, self.t1)
        graph.rm_factors(graph.ii < self.warmup - 4, store=True)

    def __call__(self):
        if not self.is_initialized:
            if self.video.counter.value != self.warmup:
                return
            self._initialize()
        elif self.t1 < self.video.counter.value:
            self._update()
        else:
            return
        self.video.update_valid_depth_mask()
