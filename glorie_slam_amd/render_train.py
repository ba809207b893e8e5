"""Training path of the renderer: `render_batch_ray` under autograd on the HIP kernels of
csrc/train.hip, and the Adam step of the mapper.

What /root/reference/src/mapper.py:390-515 (`optimizer_update_one_step`) needs from the renderer is the gradient of
a scalar loss on (depth, colour) with respect to
  * the neural-point feature tables `npc_geo_feats`, `npc_col_feats` (leaf tensors the mapper clones, mapper.py:586-611)
  * the decoder parameters (`self.decoders.parameters()`, incl. the learnable Fourier matrices)
The reference gets it from torch autograd over ~150 small ops per call; here ONE autograd.Function spans
sampling -> KNN -> decoders -> compositing: its forward is glorie_render_train_fwd + glorie_composite, its backward
glorie_composite_bwd + glorie_render_train_bwd (no torch op in between, no Python per layer).
"""
import ctypes

import torch

from . import _lib as L
from . import point_ops


class _DecoderPtrs(ctypes.Structure):
    """glorie_decoder_params / glorie_decoder_grads of include/glorie_hip.h: 52 pointers"""
    _fields_ = [("p", ctypes.c_void_p * 52)]


def decoder_tensors(decoders):
    """the decoder's tensors in the field order of glorie_decoder_params"""
    g, c = decoders.geo_decoder, decoders.color_decoder
    n = c.mlp_col_neighbor
    ts = [g.embedder._B]
    ts += [l.weight for l in g.pts_linears] + [l.bias for l in g.pts_linears]
    ts += [l.weight for l in g.fc_c] + [l.bias for l in g.fc_c]
    ts += [g.output_linear.weight, g.output_linear.bias]
    ts += [c.embedder_rel_pos._B, n.linear1.weight, n.linear1.bias, n.linear2.weight, n.linear2.bias]
    ts += [c.embedder._B, c.embedder_view_direction._B]
    ts += [l.weight for l in c.pts_linears] + [l.bias for l in c.pts_linears]
    ts += [l.weight for l in c.fc_c] + [l.bias for l in c.fc_c]
    ts += [c.output_linear.weight, c.output_linear.bias]
    assert len(ts) == 52
    return ts


_SHAPES = ([(3, 93), (32, 93), (32, 32), (32, 32), (32, 125), (32, 32)] + [(32,)] * 5 + [(32, 32)] * 5 + [(32,)] * 5 +
           [(1, 32), (1,), (3, 10), (128, 52), (128,), (32, 128), (32,), (3, 20), (3, 20),
            (128, 80), (128, 128), (128, 128), (128, 208), (128, 128)] + [(128,)] * 5 + [(128, 32)] * 5 +
           [(128,)] * 5 + [(3, 128), (3,)])


def _ptr_struct(tensors):
    st = _DecoderPtrs()
    for i, t in enumerate(tensors):
        st.p[i] = t.data_ptr() if t is not None else None
    return st


def supported(decoders):
    g, c = decoders.geo_decoder, decoders.color_decoder
    if not (g.weighting == 'distance' and c.weighting == 'distance' and g.c_dim == 32 and c.c_dim == 32 and
            c.encode_rel_pos_in_col and c.use_view_direction and c.encode_viewd and g.min_nn_num == c.min_nn_num):
        return False
    return all(tuple(t.shape) == s for t, s in zip(decoder_tensors(decoders), _SHAPES))


def _on_device(t, dev):
    """t on `dev`.  Tensors that live elsewhere - the fixed Fourier matrices of the colour decoder are plain attributes, not
    buffers, and stay on the host when the module is moved - are copied once per version: a pageable host-to-device copy in
    every forward pass stalls the host until the stream has drained (2.7 ms per mapping iteration)."""
    if t.device == dev:
        return t.detach().contiguous().float()
    # the copy rides on the tensor OBJECT the decoder holds (callers pass that object, not a detach()ed view of it): a table
    # keyed by (address, shape) hands the copy of a freed tensor to the next one the host allocator places at the same
    # address - another decoder's Fourier matrix
    hit = getattr(t, "_glorie_on_device", None)
    if hit is None or hit[0] != t._version or hit[1].device != dev:
        hit = (t._version, t.detach().contiguous().float().to(dev))
        t._glorie_on_device = hit
    return hit[1]


class RenderTrain(torch.autograd.Function):
    """(geo_feats, col_feats, *decoder tensors) -> (depth, uncertainty, colour) for samples that are already placed"""

    @staticmethod
    def forward(ctx, meta, geo_feats, col_feats, *params):
        pts, views, cloud_pos, I, w, has8, z_vals, coef, color = meta
        dev = pts.device
        Q = pts.shape[0]
        R, S = z_vals.shape
        lib = L.load()
        f32c = lambda t: t.detach().contiguous().float()
        geo_feats_c, col_feats_c = f32c(geo_feats), f32c(col_feats)
        pcs = [_on_device(p, dev) for p in params]
        ws = torch.empty(int(lib.glorie_render_train_workspace(Q)) // 4, dtype=torch.float32, device=dev)
        raw = torch.empty(Q, 4, dtype=torch.float32, device=dev)
        P = _ptr_struct(pcs)
        L.check(lib.glorie_render_train_fwd(ctypes.byref(P), L.ptr(pts), L.ptr(views), L.ptr(cloud_pos), L.ptr(geo_feats_c),
                                            L.ptr(col_feats_c), L.ptr(I), L.ptr(w), L.ptr(has8), Q, int(color),
                                            L.ptr(ws), L.ptr(raw), L.stream_ptr()), "glorie_render_train_fwd")
        depth, var, rgb, _ = point_ops.composite(raw.view(R, S, 4), z_vals, coef, return_weights=False)
        ctx.meta = meta
        ctx.saved = (geo_feats_c, col_feats_c, pcs, ws, raw)
        ctx.shapes = (geo_feats.shape, col_feats.shape)
        ctx.mark_non_differentiable(var)
        return depth, var, rgb

    @staticmethod
    def backward(ctx, g_depth, g_var, g_rgb):
        pts, views, cloud_pos, I, w, has8, z_vals, coef, color = ctx.meta
        geo_feats_c, col_feats_c, pcs, ws, raw = ctx.saved
        dev = pts.device
        Q = pts.shape[0]
        R, S = z_vals.shape
        lib = L.load()
        gd = g_depth.contiguous().float() if g_depth is not None else None
        gc = g_rgb.contiguous().float() if g_rgb is not None else None
        d_raw = torch.empty(Q, 4, dtype=torch.float32, device=dev)
        L.check(lib.glorie_composite_bwd(L.ptr(raw), L.ptr(z_vals), R, S, float(coef), L.ptr(gd), L.ptr(gc), L.ptr(d_raw),
                                         L.stream_ptr()), "glorie_composite_bwd")
        need = ctx.needs_input_grad            # (meta, geo_feats, col_feats, *params)
        d_geo = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=dev) if need[1] else None
        # the colour table's gradient buffer is needed by the kernel whenever the colour stage runs
        d_col = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=dev) if (need[2] or color) else None
        # one zero-filled buffer for all parameter gradients (views of it are returned): 1 fill instead of 50
        wanted = [(need[3 + i] and i not in (28, 29)) for i in range(len(pcs))]
        offs, total = [], 0
        for i, p in enumerate(pcs):
            offs.append(total)
            if wanted[i]:
                total += (p.numel() + 3) // 4 * 4                   # keep every view 16-byte aligned
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grads = [flat[offs[i]:offs[i] + p.numel()].view(p.shape) if wanted[i] else None for i, p in enumerate(pcs)]
        P, G = _ptr_struct(pcs), _ptr_struct(grads)
        L.check(lib.glorie_render_train_bwd(ctypes.byref(P), ctypes.byref(G), L.ptr(pts), L.ptr(views), L.ptr(cloud_pos),
                                            L.ptr(geo_feats_c), L.ptr(col_feats_c), L.ptr(I), L.ptr(w), L.ptr(has8), Q,
                                            int(color), L.ptr(ws), L.ptr(d_raw), L.ptr(d_geo), L.ptr(d_col),
                                            L.stream_ptr()), "glorie_render_train_bwd")
        ctx.saved = None
        return (None, d_geo, d_col if need[2] else None) + tuple(grads)


def render_rays(renderer, npc, decoders, rays_d, rays_o, stage, gt_depth, npc_geo_feats, npc_col_feats, cloud_pos,
                dynamic_r_query):
    """render_batch_ray for a batch whose rays all carry a depth prior, differentiable with respect to the feature
    tables and the decoder parameters.  Returns None if the batch holds a ray without depth (general path)."""
    S = renderer.N_surface
    R = rays_o.shape[0]
    g = decoders.geo_decoder
    rad = dynamic_r_query if renderer.use_dynamic_radius else None
    with torch.no_grad():
        z_vals, pts, views, rq, n_zero = point_ops.ray_samples(rays_o, rays_d, gt_depth, rad, S,
                                                               renderer.near_end_surface, renderer.far_end_surface)
        # neighbours, IDW weights and mask from one launch, the search bounded by the query radius (point_ops.KnnIndex.search)
        D, I, nn_num, w, has8 = npc.find_neighbors_faiss(pts, step='query', dynamic_radius=rq,
                                                         weights=(g.min_nn_num, False, True))
        counts, valid = point_ops.ray_counts(has8, S, 3)
        # the zero-depth count travels to pinned memory behind the sampling kernel and is looked at after the forward pass
        # has been enqueued: `int(n_zero)` here would stall the host until the previous iteration's backward and Adam
        # step have drained (the loop was host-bound on exactly that wait)
        # (renderer.assume_depth: the caller has checked that every ray it will draw carries a depth prior - the mapping
        # loop does so once per keyframe; nothing of the iteration then needs the host, and it can be recorded into a hipGraph)
        checked = not getattr(renderer, "assume_depth", False)
        if checked:
            flag = renderer._pinned_flag()
            flag.copy_(n_zero, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
    cp = (cloud_pos if cloud_pos is not None else npc.cloud_pos()).detach().contiguous().float()
    meta = (pts, views, cp, I.contiguous(), w.contiguous(), has8.contiguous(), z_vals, renderer.sigmoid_coefficient,
            stage == "color")
    depth, var, rgb = RenderTrain.apply(meta, npc_geo_feats, npc_col_feats, *decoder_tensors(decoders))
    if checked:
        ev.synchronize()
        if int(flag[0]) != 0:
            return None                  # a ray without depth: the general path redoes the batch
    return depth, var, rgb, valid, counts


class FeatureAdam:
    """torch.optim.Adam semantics (mapper.py:612-624) on glorie_adam_step: one launch per tensor, optional row mask
    for the frustum-selected rows of a feature table (rows outside it keep their value AND their moments)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        """capturable: the step count lives in device memory (one word for all tensors: they start together) and is advanced
        on the stream (glorie_counter_add, glorie_adam_step_dev / glorie_adam_multi_dev) - `step()` can then be recorded
        into a hipGraph with the backward pass in front of it; after replays the host side is brought up to date with
        `advance(n)`"""
        self.capturable = bool(capturable)
        self._step_dev = None
        self.param_groups = []
        for g in params:
            g = dict(g) if isinstance(g, dict) else {"params": list(g)}
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            g["params"] = list(g["params"])
            self.param_groups.append(g)
        self.state = {}

    def zero_grad(self):
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    def _upload(self, buf, dev):
        """the pointer table -> device without stalling the host: the gradient buffers move almost every iteration, and a
        `.to(device)` from pageable memory blocks until the stream has drained - i.e. until the backward pass that was just
        enqueued is through (1.2 ms of a 4.2 ms mapping iteration, tools/prof_sequence.py).  Four pinned staging buffers
        in turn, each guarded by the event of the copy that last read it; the device copy is stream-ordered behind the
        previous step's kernel, so one device buffer is enough."""
        n = len(buf)
        ring = getattr(self, "_ring", None)
        if ring is None or ring["pin"][0].numel() < n or ring["dev"].device != dev:
            ring = self._ring = {"pin": [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)],
                                 "ev": [None] * 4, "dev": torch.empty(n, dtype=torch.uint8, device=dev), "i": 0}
        i = ring["i"] = (ring["i"] + 1) % 4
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()
        pin = ring["pin"][i]
        pin[:n].copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
        ring["dev"][:n].copy_(pin[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        ring["ev"][i] = ev
        return ring["dev"]

    MULTI_MAX = 1 << 16           # tensors up to this many elements share one launch (the decoder's 52 tensors)

    def capturable_fallback(self):
        """leave the capturable mode (a recording failed): later steps pass the host's step count by value again"""
        self.capturable = False

    def advance(self, n):
        """host bookkeeping for `n` replays of a recorded step: step counts and the parameters' version counters"""
        for g in self.param_groups:
            for p in g["params"]:
                st = self.state.get(id(p))
                if st is not None:
                    st["step"] += int(n)
                    torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def step(self, row_masks=None):
        import struct
        lib = L.load()
        small = []
        sdev = None
        if self.capturable:
            dev0 = next(p.device for g in self.param_groups for p in g["params"])
            if self._step_dev is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FeatureAdam(capturable): take one eager step before recording")
                self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev0)
            sdev = self._step_dev
            L.check(lib.glorie_counter_add(L.ptr(sdev), 1, L.stream_ptr(dev0)), "glorie_counter_add")
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(id(p))
                if st is None:               # (setdefault would build - and zero-fill - the default on every call)
                    st = self.state[id(p)] = {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}
                st["step"] += 1
                # the kernels write through raw pointers: tell autograd (and every cache keyed on `_version`, e.g. the packed
                # parameter buffer of the inference kernels, POINT._packed) that the tensor changed
                torch.autograd.graph.increment_version(p)
                mask = row_masks.get(id(p)) if row_masks else None
                if not (p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32):
                    raise RuntimeError("FeatureAdam: contiguous float32 parameters expected")
                if mask is None and p.numel() <= self.MULTI_MAX and p.is_cuda:
                    small.append((p, st, g))
                    continue
                row_len = p.shape[-1] if mask is not None else 1
                m8 = mask.to(torch.uint8).contiguous() if mask is not None else None
                if sdev is not None:
                    L.check(lib.glorie_adam_step_dev(L.ptr(p), L.ptr(p.grad), L.ptr(st["m"]), L.ptr(st["v"]), p.numel(),
                                                     float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                                     float(g["eps"]), L.ptr(sdev), L.ptr(m8), int(row_len),
                                                     L.stream_ptr(p.device)), "glorie_adam_step_dev")
                    continue
                L.check(lib.glorie_adam_step(L.ptr(p), L.ptr(p.grad), L.ptr(st["m"]), L.ptr(st["v"]), p.numel(),
                                             float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                             int(st["step"]), L.ptr(m8), int(row_len), L.stream_ptr(p.device)),
                        "glorie_adam_step")
        if sdev is not None and len({st["step"] for g in self.param_groups for p in g["params"]
                                      for st in [self.state.get(id(p))] if st is not None}) > 1:
            raise RuntimeError("FeatureAdam(capturable): every tensor must be at the same step")
        if small and len({st["step"] for _, st, _ in small}) > 1:
            for p, st, g in small:                                  # tensors at different steps: one launch each
                L.check(lib.glorie_adam_step(L.ptr(p), L.ptr(p.grad), L.ptr(st["m"]), L.ptr(st["v"]), p.numel(),
                                             float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                             int(st["step"]), None, 1, L.stream_ptr(p.device)), "glorie_adam_step")
        elif small:
            # glorie_adam_multi: one 80-byte table entry per tensor, re-packed and re-sent only when something it holds changed.
            # The gradients of a backward pass are views of ONE buffer (RenderTrain.backward) that the caching allocator places
            # somewhere else almost every iteration: the table then holds their byte offsets and the buffer's address travels
            # as a kernel argument - with absolute pointers the table was rebuilt and uploaded on 1206 of 1240 steps.
            # The key covers EVERYTHING an entry holds: the mapper rewrites param_groups[i]['lr'] between its stages
            # (mapper.py:412-414).
            dev = small[0][0].device
            gbase = small[0][0].grad.untyped_storage().data_ptr()
            shared = all(p.grad.untyped_storage().data_ptr() == gbase for p, _, _ in small)
            goff = (lambda p: p.grad.data_ptr() - gbase) if shared else (lambda p: p.grad.data_ptr())
            key = tuple((p.data_ptr(), goff(p), st["m"].data_ptr(), st["v"].data_ptr(), p.numel(), float(g["lr"]),
                         float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for p, st, g in small) + (shared,)
            if getattr(self, "_table_key", None) != key:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FeatureAdam: the pointer table changed under capture (take an eager step with the "
                                       "same tensors first)")
                buf = b"".join(struct.pack("<qqqqqffffqqq", p.data_ptr(), goff(p), st["m"].data_ptr(),
                                           st["v"].data_ptr(), p.numel(), float(g["lr"]), float(g["betas"][0]),
                                           float(g["betas"][1]), float(g["eps"]), 0, 0, 0) for p, st, g in small)
                self._table = self._upload(buf, dev)
                self._table_key = key
                self._table_max = max(p.numel() for p, _, _ in small)
            if sdev is not None:
                L.check(lib.glorie_adam_multi_dev(L.ptr(self._table), len(small), self._table_max, L.ptr(sdev),
                                                  gbase if shared else None, L.stream_ptr(dev)), "glorie_adam_multi_dev")
            else:
                L.check(lib.glorie_adam_multi(L.ptr(self._table), len(small), self._table_max, int(small[0][1]["step"]),
                                              gbase if shared else None, L.stream_ptr(dev)), "glorie_adam_multi")
