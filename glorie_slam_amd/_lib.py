"""ctypes binding of libglorie_hip.so (the C ABI declared in include/glorie_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails the
caller gets an exception.  Only plumbing lives here -- device pointers come from
``tensor.data_ptr()`` and the stream from ``torch.cuda.current_stream()``.
"""
import ctypes
import os
import threading

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libglorie_hip.so")

GLORIE_F16, GLORIE_F32 = 0, 1
_STATUS = {0: "GLORIE_OK", -1: "GLORIE_EINVAL", -2: "GLORIE_EHIP", -3: "GLORIE_ENOMEM",
           -4: "GLORIE_EUNSUPPORTED"}

_c_int, _c_f, _vp, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/glorie_hip.h one to one
SIGNATURES = {
    "glorie_version": (ctypes.c_char_p, []),
    "glorie_last_hip_error": (_c_int, []),
    "glorie_ctx_create": (_c_int, [ctypes.POINTER(_vp), _sz]),
    "glorie_ctx_destroy": (_c_int, [_vp]),
    "glorie_ctx_reserve": (_c_int, [_vp, _sz]),
    "glorie_ctx_generation": (ctypes.c_ulonglong, [_vp]),
    "glorie_corr_index_fwd": (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "glorie_corr_lookup_pyramid": (_c_int, [_vp, _c_int, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "glorie_corr_otf": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _vp] + [_c_int] * 4 + [_vp]),
    "glorie_corr_otf_encode": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _vp] + [_c_int] * 4 + [_vp, _vp, _vp, _c_int, _vp]),
    "glorie_altcorr_fwd": (_c_int, [_vp] * 4 + [_c_int] * 8 + [_vp]),
    "glorie_bias_act": (_c_int, [_vp, _c_int, _vp, _vp, _c_int, ctypes.c_long, _c_int, _c_int, _vp]),
    "glorie_gru_glo_terms": (_c_int, [_vp, _c_int, _vp, _vp, _c_int, _vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _vp]),
    "glorie_gru_gate_zr": (_c_int, [_vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _vp]),
    "glorie_gru_gate_q": (_c_int, [_vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _vp]),
    "glorie_segment_mean": (_c_int, [_vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _vp]),
    "glorie_conv3x3_small": (_c_int, [_vp, _c_int, _vp, _c_int, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_float, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "glorie_gru_glo_from_tiles": (_c_int, [_vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _vp]),
    "glorie_update_bookkeeping": (_c_int, [_vp, _vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_f, _vp, _c_int,
                                           _vp]),
    "glorie_conv_igemm_heads": (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _vp, _vp,
                                         _c_int, _c_int, _c_int, _c_int, _vp]),
    "glorie_conv_upsample": (_c_int, [_vp, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "glorie_conv_stencil": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, ctypes.c_float, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "glorie_conv_igemm": (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _vp,
                                   _c_int, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp, _c_int, _vp,
                                   _c_int, _c_int, _c_int, _vp]),
    "glorie_flow_conv7": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "glorie_motion": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_float, _vp]),
    "glorie_motion_padded": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_float, _vp]),
    "glorie_flow_pad": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "glorie_flow_conv7_padded": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "glorie_publish_flag": (_c_int, [_vp, _vp, _vp, _vp]),
    "glorie_dspo_prepare": (_c_int, [_vp] * 4 + [_c_int] * 4 + [ctypes.c_float, _c_int, ctypes.c_float, _vp, _vp, _c_int] + [_vp] * 9),
    "glorie_corr_lookup_pyramid_tiled": (_c_int, [_vp, _c_int, _vp, _vp] + [_c_int] * 5 + [_vp]),
    "glorie_corr_build": (_c_int, [_vp] * 5 + [_c_int] * 5 + [_vp]),
    "glorie_corr_lookup_arena": (_c_int, [_vp, _c_int, _vp, _vp, _vp] + [_c_int] * 5 + [_vp]),
    "glorie_corr_lookup_tiled_cl": (_c_int, [_vp, _c_int, _vp, _vp, _c_int, _vp] + [_c_int] * 5 + [_vp]),
    "glorie_corr_dm_level_halfs": (ctypes.c_long, [_c_int, _c_int, _c_int]),
    "glorie_corr_dm_build": (_c_int, [_vp] * 5 + [_c_int] * 5 + [_vp]),
    "glorie_corr_dm_lookup": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp]),
    "glorie_valid_depth_mask": (_c_int, [_vp] * 4 + [_c_int] * 4 + [ctypes.c_float, _c_int, _vp, _vp, _vp]),
    "glorie_reproject": (_c_int, [_vp] * 7 + [_c_int] * 3 + [_vp]),
    "glorie_reproject_motion": (_c_int, [_vp] * 9 + [_c_int] * 3 + [ctypes.c_float, _vp]),
    "glorie_frame_distance": (_c_int, [_vp] * 6 + [_c_int] * 3 + [_c_f, _vp]),
    "glorie_iproj": (_c_int, [_vp] * 4 + [_c_int] * 3 + [_vp]),
    "glorie_depth_filter": (_c_int, [_vp] * 6 + [_c_int] * 4 + [_vp]),
    "glorie_cvx_upsample": (_c_int, [_vp] * 4 + [_c_int] * 5 + [_vp]),
    "glorie_cvx_upsample_nhwc": (_c_int, [_vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "glorie_ba": (_c_int, [_vp] * 10 + [_c_int] * 8 + [_c_f, _c_f, _c_int, _c_int, _vp, _vp, _vp]),
    "glorie_ba_build_system": (_c_int, [_vp] * 10 + [_c_int] * 8 + [_vp, _vp]),
    "glorie_ba_solve_update": (_c_int, [_vp] * 5 + [_c_int] * 7 + [_c_f, _c_f, _c_int, _c_int] + [_vp] * 4),
    "glorie_ba_pack_system": (_c_int, [_vp, _vp, _c_int, _c_int, _vp]),
    "glorie_comm_unique_id": (_c_int, [_vp]),
    "glorie_comm_init": (_c_int, [_vp, _vp, _c_int, _c_int]),
    "glorie_comm_destroy": (_c_int, [_vp]),
    "glorie_comm_world": (_c_int, [_vp]),
    "glorie_allreduce_normal_eq": (_c_int, [_vp, _vp, _sz, _vp]),
    "glorie_allgather_rows": (_c_int, [_vp, _vp, _vp, _sz, _vp]),
    "glorie_dspo_scale_shift": (_c_int, [_vp] * 14 + [_c_int] * 6 + [_c_f, _c_f, _c_f, _vp, _vp]),
    "glorie_knn_build": (_c_int, [_vp, _vp, _c_int, _c_f, _c_int, _vp, _vp, _vp, _vp]),
    "glorie_knn_query": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_f, _vp, _vp, _vp, _vp, _vp]),
    "glorie_idw_gather": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_f, _vp, _c_int,
                                   _c_int, _vp, _vp, _vp, _vp]),
    "glorie_knn_query_image": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_f, _vp, _vp, _vp, _vp, _c_int, _c_int,
                                        _vp]),
    "glorie_knn_query_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_f, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int,
                                          _c_int, _vp, _vp, _vp]),
    "glorie_idw_gather2": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_f, _vp, _c_int,
                                    _c_int, _vp, _vp, _vp, _vp, _vp]),
    "glorie_decoder_pack_floats": (_sz, []),
    "glorie_render_mlp": (_c_int, [_vp] * 10 + [_c_int, _vp, _vp, _c_int, _vp, _vp]),
    "glorie_composite": (_c_int, [_vp, _vp, _c_int, _c_int, _c_f, _vp, _vp, _vp, _vp, _vp]),
    "glorie_ray_samples": (_c_int, [_vp] * 5 + [_c_int, _c_int, _c_f, _c_f] + [_vp] * 6),
    "glorie_ray_samples_camera": (_c_int, [_vp, _c_int, ctypes.c_long] + [_vp] * 3 + [_c_int, _c_int, _c_f, _c_f] + [_vp] * 6),
    "glorie_proj_depth": (_c_int, [_vp, _vp, ctypes.c_long, _vp, _c_f, _c_f, _c_f, _c_f, _c_int, _c_int, _vp, _vp]),
    "glorie_ray_counts": (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp]),
    "glorie_render_train_workspace": (_sz, [ctypes.c_long]),
    "glorie_render_train_fwd": (_c_int, [_vp] * 9 + [ctypes.c_long, _c_int, _vp, _vp, _vp]),
    "glorie_render_train_bwd": (_c_int, [_vp] * 10 + [ctypes.c_long, _c_int, _vp, _vp, _vp, _vp, _vp]),
    "glorie_composite_bwd": (_c_int, [_vp, _vp, _c_int, _c_int, _c_f, _vp, _vp, _vp, _vp]),
    "glorie_adam_step": (_c_int, [_vp, _vp, _vp, _vp, ctypes.c_long, _c_f, _c_f, _c_f, _c_f, _c_int, _vp, _c_int, _vp]),
    "glorie_adam_multi": (_c_int, [_vp, _c_int, ctypes.c_long, _c_int, _vp, _vp]),
    "glorie_adam_step_dev": (_c_int, [_vp, _vp, _vp, _vp, ctypes.c_long, _c_f, _c_f, _c_f, _c_f, _vp, _vp, _c_int, _vp]),
    "glorie_adam_multi_dev": (_c_int, [_vp, _c_int, ctypes.c_long, _vp, _vp, _vp]),
    "glorie_counter_add": (_c_int, [_vp, _c_int, _vp]),
    "glorie_ba_status": (_c_int, [_vp, ctypes.POINTER(_c_int), _vp]),
    "glorie_ba_set_gate": (_c_int, [_vp, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


class GlorieError(RuntimeError):
    pass


def load():
    """dlopen the library (once).  Raises if it has not been built -- never falls back."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise GlorieError(
                    f"{LIB_PATH} is missing: build it with `python glorie_slam_amd/build.py` "
                    "(or __graft_entry__.build()); there is no CPU fallback for the hot path")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(status, what):
    if status != 0:
        lib = load()
        raise GlorieError(f"{what} failed: {_STATUS.get(status, status)} "
                          f"(hipError {lib.glorie_last_hip_error()})")


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


BA_TARGETS_HWC = 2   # GLORIE_BA_TARGETS_HWC (include/glorie_hip.h)


def need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GlorieError("glorie_hip kernels take device tensors; got a CPU tensor "
                              "(there is no CPU fallback on the product path)")


def need_contiguous(**named):
    # same contract as CHECK_CONTIGUOUS in the reference binding (src/lib/droid.cpp:85-86)
    for k, t in named.items():
        if t is not None and not t.is_contiguous():
            raise RuntimeError(f"{k} must be contiguous")


def dtype_code(t):
    if t.dtype == torch.float16:
        return GLORIE_F16
    if t.dtype == torch.float32:
        return GLORIE_F32
    raise GlorieError(f"unsupported dtype {t.dtype}")


class Context:
    """Per-process/GPU scratch arena (glorie_ctx)."""

    def __init__(self, scratch_bytes=64 << 20):
        lib = load()
        h = ctypes.c_void_p()
        check(lib.glorie_ctx_create(ctypes.byref(h), scratch_bytes), "glorie_ctx_create")
        self.handle = h

    def generation(self):
        """how often the scratch arena moved (launches recorded before a move point into freed memory)"""
        return int(load().glorie_ctx_generation(self.handle))

    def reserve(self, scratch_bytes):
        check(load().glorie_ctx_reserve(self.handle, scratch_bytes), "glorie_ctx_reserve")

    def ba_status(self):
        out = (ctypes.c_int * 4)()
        check(load().glorie_ba_status(self.handle, out, stream_ptr()), "glorie_ba_status")
        return list(out)

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.glorie_ctx_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


_default_ctx = {}


def default_context():
    dev = torch.cuda.current_device()
    if dev not in _default_ctx:
        _default_ctx[dev] = Context()
    return _default_ctx[dev]
