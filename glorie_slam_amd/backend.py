"""Global / loop bundle adjustment driver (scope row N3) -- same interface as
/root/reference/src/backend.py (`Backend.ba`, `.dense_ba`, `.loop_ba`).

Both entry points build a throw-away FactorGraph with on-the-fly correlation (`corr_impl='alt'`), let
`add_backend_proximity_factors` choose the edges and run `update_lowmem`.  dense_ba optimises every keyframe
(max (radius + 2) * 2 edges per frame, backend.py:50-70); loop_ba seeds the graph with the frontend's local
edges and only adds covisible pairs that close a loop into the last `loop_window` frames (backend.py:74-98).
"""
import copy

import torch

from .factor_graph import FactorGraph


class Backend:
    def __init__(self, net, video, cfg):
        trk = cfg['tracking']
        be = trk['backend']
        self.video = video
        self.update_op = net.update
        self.device = cfg['device']
        self.t0 = self.t1 = 0                 # global optimisation window (kept for interface parity)
        self.beta = trk['beta']
        self.backend_thresh, self.backend_radius = be['thresh'], be['radius']
        self.backend_nms, self.backend_normalize = be['nms'], be['normalize']
        self.backend_loop_window, self.backend_loop_thresh = be['loop_window'], be['loop_thresh']
        self.backend_loop_radius, self.backend_loop_nms = be['loop_radius'], be['loop_nms']
        self.output = f"{cfg['data']['output']}/{cfg['setting']}/{cfg['scene']}" if 'data' in cfg else ''

    def _graph(self, max_factors):
        return FactorGraph(self.video, self.update_op, device=self.device, corr_impl='alt',
                           max_factors=max_factors)

    @torch.no_grad()
    def ba(self, t_start, t_end, steps, graph, nms, radius, thresh, max_factors, t_start_loop=None,
           loop=False, motion_only=False, enable_wq=True):
        """edge selection + `steps` low-memory updates; poses up to t_start_loop stay fixed.
        Returns the number of edges that were added (0: nothing to optimise)."""
        if not loop or t_start_loop is None:
            t_start_loop = t_start
        if t_start_loop < t_start:
            raise AssertionError(f'short: {t_start_loop}, long: {t_start}.')
        n_edges = graph.add_backend_proximity_factors(t_start, t_end, nms, radius, thresh, max_factors,
                                                      self.beta, t_start_loop, loop)
        if n_edges > 0:
            # anchor the first frame of the (loop) window, not t_start: this is what prevents drift
            graph.update_lowmem(t0=t_start_loop + 1, t1=t_end, itrs=2, use_inactive=False, steps=steps,
                                enable_wq=enable_wq)
        graph.clear_edges()
        return n_edges

    @torch.no_grad()
    def dense_ba(self, steps=6, enable_wq=True):
        t_start, t_end = 0, self.video.counter.value
        n = t_end - t_start
        budget = 2 * (self.backend_radius + 2) * n
        if self.backend_normalize:
            self.video.normalize()
        graph = self._graph(budget)
        n_edges = self.ba(t_start, t_end, steps, graph, self.backend_nms, self.backend_radius,
                          self.backend_thresh, budget, motion_only=False, enable_wq=enable_wq)
        del graph
        self.video.set_dirty(t_start, t_end)
        self.video.update_valid_depth_mask()
        return n, n_edges

    @torch.no_grad()
    def loop_ba(self, t_start, t_end, steps=6, motion_only=False, local_graph=None, enable_wq=True):
        window = self.backend_loop_window
        budget = 8 * window
        t_start_loop = max(t_end - window, 0)
        graph = self._graph(budget)
        if local_graph is not None:           # start from the frontend's edges and their state
            for name in ('ii', 'jj', 'age', 'net', 'target', 'weight'):
                value = getattr(local_graph, name)
                if value is not None:
                    setattr(graph, name, copy.deepcopy(value))
            graph._topo += 1
        n_edges = self.ba(t_start, t_end, steps, graph, self.backend_loop_nms, self.backend_loop_radius,
                          self.backend_loop_thresh, budget - len(graph.ii), t_start_loop=t_start_loop,
                          loop=True, motion_only=motion_only, enable_wq=enable_wq)
        del graph
        return t_end - t_start_loop, n_edges
