"""ctypes binding of ``libkfusion_hip.so`` (the C ABI declared in ``include/kfusion_hip.h``).

There is no CPU fallback: if the shared library is missing or cannot be loaded this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkfusion_hip.so")

_lib = None

vp, u, i32, f, u64, i64 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float, C.c_uint64, C.c_int64
fp, u32p, i32p, u16p, u8p, i16p, cstr = (C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_uint16), C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.c_char_p)

# name -> (restype, argtypes).  Device pointers are passed as c_void_p integers.
SIGNATURES = {
    "kfh_init": (i32, []),
    "kfh_clean": (None, []),
    "kfh_device_count": (i32, []),
    "kfh_last_error": (cstr, []),
    "kfh_version": (cstr, []),
    "kfh_create": (vp, [u, u, u32p, fp, fp, i32p, i32, i32, vp]),
    "kfh_create_with_pose": (vp, [u, u, u32p, fp, fp, i32p, i32, i32, vp]),
    "kfh_destroy": (None, [vp]),
    "kfh_reset": (None, [vp]),
    "kfh_synchronise": (None, [vp]),
    "kfh_synchronise_all": (None, []),
    "kfh_preprocessing": (i32, [vp, u16p, u, u]),
    "kfh_preprocessing_device": (i32, [vp, vp, u, u]),
    "kfh_tracking": (i32, [vp, fp, f, u, u]),
    "kfh_integration": (i32, [vp, fp, u, f, u]),
    "kfh_raycasting": (i32, [vp, fp, f, u]),
    "kfh_compute_frame": (i32, [vp, vp, i32, u, u, fp, u, u, f, f, u]),
    "kfh_render_depth": (None, [vp, u8p, u, u]),
    "kfh_render_track": (None, [vp, u8p, u, u]),
    "kfh_render_volume": (None, [vp, u8p, u, u, i32, i32, fp, f]),
    "kfh_dump_volume": (i32, [vp, cstr]),
    "kfh_get_pose": (None, [vp, fp]),
    "kfh_set_pose": (None, [vp, fp]),
    "kfh_get_position": (None, [vp, fp]),
    "kfh_get_tracked": (i32, [vp]),
    "kfh_get_integrated": (i32, [vp]),
    "kfh_get_error": (i32, [vp]),
    "kfh_set_view_pose": (None, [vp, fp]),
    "kfh_get_model_dimensions": (None, [vp, fp]),
    "kfh_get_model_resolution": (None, [vp, u32p]),
    "kfh_get_computation_resolution": (None, [vp, u32p]),
    "kfh_read_frame_log": (i32, [vp, u, fp, i32p, i32p]),
    "kfh_icp_iterations": (u64, [vp]),
    "kfh_icp_launches_per_frame": (u, [vp]),
    "kfh_icp_level_iterations": (None, [vp, C.POINTER(C.c_uint64)]),
    "kfh_integrate_updated": (u64, [vp]),
    "kfh_solve_fallbacks": (u64, [vp]),
    "kfh_barrier_timeouts": (u64, [vp]),
    "kfh_set_profiling": (None, [vp, u]),
    "kfh_set_profiling_stride": (None, [vp, u]),
    "kfh_set_profiling_rotation": (None, [vp, i32]),
    "kfh_read_profile": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "kfh_read_buffer": (i64, [vp, cstr, vp, u64]),
    "kfh_write_buffer": (i64, [vp, cstr, vp, u64]),
    "kfh_device_pointer": (vp, [vp, cstr]),
    "kfh_k_init_volume": (None, [vp, u32p, vp]),
    "kfh_k_mm2meters": (i32, [vp, u, u, vp, u, u, vp]),
    "kfh_k_bilateral_filter": (None, [vp, vp, u, u, fp, f, i32, vp]),
    "kfh_k_preprocess": (i32, [vp, vp, u, u, vp, u, u, fp, f, vp]),
    "kfh_k_half_sample": (None, [vp, vp, u, u, f, i32, vp]),
    "kfh_k_depth2vertex": (None, [vp, vp, u, u, fp, vp]),
    "kfh_k_vertex2normal": (None, [vp, vp, u, u, vp]),
    "kfh_k_vertex_normal": (None, [vp, vp, vp, u, u, fp, vp]),
    "kfh_k_track": (None, [vp, vp, vp, u, u, vp, vp, u, u, fp, fp, f, f, i32, vp]),
    "kfh_k_reduce": (None, [vp, vp, u, u, u, u, vp]),
    "kfh_k_track_reduce": (None, [vp, vp, vp, vp, u, u, vp, vp, u, u, fp, fp, f, f, i32, vp]),
    "kfh_k_update_pose": (i32, [vp, vp, f, vp]),
    "kfh_k_check_pose": (i32, [vp, vp, vp, u, u, f, vp]),
    "kfh_k_integrate": (None, [vp, u32p, fp, vp, u, u, fp, fp, f, f, vp]),
    "kfh_k_raycast": (None, [vp, vp, u, u, vp, u32p, fp, fp, f, f, f, f, vp]),
    "kfh_k_render_depth": (None, [vp, vp, u, u, f, f, vp]),
    "kfh_k_render_track": (None, [vp, vp, u, u, vp]),
    "kfh_k_render_volume": (None, [vp, u, u, vp, u32p, fp, fp, f, f, f, f, fp, fp, vp]),
}


def lib():
    """Load the HIP backend; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP backend missing: %s not built (run `python -m slambench_amd.build`)" % LIB_PATH)
    # One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64 under the SONAMEs of
    # /opt/rocm's; whichever is loaded first serves every later request for that SONAME made through the loader's search,
    # but torch loads its copies by path.  If this library came first the process would hold two runtimes, and device
    # pointers made by one (torch tensors) are strangers to the other (measured: kernels launched on them silently do
    # nothing).  So when torch is installed it is imported before the backend; KFH_NO_TORCH_PRELOAD=1 skips that for
    # processes that never touch torch.
    if "torch" not in sys.modules and os.environ.get("KFH_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def f32(values):
    return (C.c_float * len(values))(*[float(v) for v in values])


def u32(values):
    return (C.c_uint32 * len(values))(*[int(v) for v in values])


def i32a(values):
    return (C.c_int32 * len(values))(*[int(v) for v in values])
