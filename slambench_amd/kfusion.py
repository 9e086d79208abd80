"""Python mirror of the reference's ``class Kfusion`` (kfusion/include/kernels.h:83-195) over the C ABI.

Same method names, argument meaning and return values as the reference interface, so parity tests
read like the reference's own driver (kfusion/src/benchmark.cpp:114-179).  All compute happens in
``libkfusion_hip.so``; this file only marshals arguments.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi

# kfusion/include/default_parameters.h:25-43
DEFAULT_PYRAMID = (10, 5, 4)
DEFAULT_MU = 0.1
DEFAULT_ICP_THRESHOLD = 1e-5
DEFAULT_INTEGRATION_RATE = 2
DEFAULT_RENDERING_RATE = 4
DEFAULT_TRACKING_RATE = 1


class Kfusion:
    def __init__(self, input_size, volume_resolution, volume_dimensions, init_pose, pyramid=DEFAULT_PYRAMID,
                 device=-1, stream=None):
        self.L = _capi.lib()
        self.W, self.H = int(input_size[0]), int(input_size[1])
        self.volume_resolution = tuple(int(v) for v in volume_resolution)
        self.volume_dimensions = tuple(float(v) for v in volume_dimensions)
        self.pyramid = list(pyramid)
        init_pose = np.asarray(init_pose, np.float32)
        args = (self.W, self.H, _capi.u32(self.volume_resolution), _capi.f32(self.volume_dimensions))
        tail = (_capi.i32a(self.pyramid), len(self.pyramid), int(device), stream)
        if init_pose.shape == (4, 4):
            self.h = self.L.kfh_create_with_pose(*args, _capi.f32(init_pose.reshape(-1)), *tail)
        else:
            self.h = self.L.kfh_create(*args, _capi.f32(init_pose.reshape(-1)[:3]), *tail)
        if not self.h:
            raise RuntimeError("kfh_create failed: %s" % self.L.kfh_last_error().decode())

    # ---- lifecycle
    def close(self):
        if getattr(self, "h", None):
            self.L.kfh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.L.kfh_reset(self.h)

    def synchronise(self):
        self.L.kfh_synchronise(self.h)

    # ---- pipeline (kernels.h:165-178)
    def preprocessing(self, input_depth, input_size=None):
        """input_depth: host uint16 array (H, W) in millimetres, or an int device pointer with input_size."""
        if isinstance(input_depth, int):
            r = self.L.kfh_preprocessing_device(self.h, input_depth, int(input_size[0]), int(input_size[1]))
        else:
            d = np.ascontiguousarray(input_depth, np.uint16)
            r = self.L.kfh_preprocessing(self.h, d.ctypes.data_as(C.POINTER(C.c_uint16)), d.shape[1], d.shape[0])
        if r < 0:
            raise ValueError("Invalid ratio." if r == -1 else "kfh_preprocessing failed: %d" % r)
        return True

    def _verdict(self, r, what):
        """1 / 0 of the reference's bool returns; a negative value is a device-side failure (kfusion_hip.h, kfh_tracking)."""
        if r < 0:
            raise RuntimeError("%s failed (%d): %s" % (what, r, self.L.kfh_last_error().decode()))
        return bool(r)

    def tracking(self, k, icp_threshold, tracking_rate, frame):
        return self._verdict(self.L.kfh_tracking(self.h, _capi.f32(k), icp_threshold, tracking_rate, frame), "kfh_tracking")

    def integration(self, k, integration_rate, mu, frame):
        return self._verdict(self.L.kfh_integration(self.h, _capi.f32(k), integration_rate, mu, frame), "kfh_integration")

    def raycasting(self, k, mu, frame):
        return self._verdict(self.L.kfh_raycasting(self.h, _capi.f32(k), mu, frame), "kfh_raycasting")

    def get_error(self):
        """KFH_OK (0) or the sticky error code of the context; synchronises."""
        return int(self.L.kfh_get_error(self.h))

    def compute_frame(self, input_depth, input_size, k, integration_rate, tracking_rate, icp_threshold, mu, frame):
        """Kfusion::computeFrame (cpp/kernels.cpp:1048-1055); queues the frame without a host round trip."""
        if isinstance(input_depth, int):
            ptr, is_dev = input_depth, 1
        else:
            d = np.ascontiguousarray(input_depth, np.uint16)
            self._keep = d
            ptr, is_dev = d.ctypes.data, 0
        r = self.L.kfh_compute_frame(self.h, ptr, is_dev, int(input_size[0]), int(input_size[1]), _capi.f32(k),
                                     integration_rate, tracking_rate, icp_threshold, mu, frame)
        if r == -1:
            raise ValueError("Invalid ratio.")
        self._verdict(r, "kfh_compute_frame")

    def _render(self, fn, *extra):
        out = np.zeros((self.H, self.W, 4), np.uint8)
        fn(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8)), self.W, self.H, *extra)
        self.synchronise()
        return out

    def render_depth(self):
        return self._render(self.L.kfh_render_depth)

    def render_track(self):
        return self._render(self.L.kfh_render_track)

    def render_volume(self, frame, rate, k, largestep, out=None):
        buf = np.zeros((self.H, self.W, 4), np.uint8) if out is None else out
        self.L.kfh_render_volume(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), self.W, self.H, frame, rate,
                                 _capi.f32(k), largestep)
        self.synchronise()
        return buf

    def queue_renders(self, depth_out, track_out, volume_out, frame, rate, k, largestep):
        """The three renders of the frame just queued, without waiting for them: `*_out` are (H, W, 4) uint8 arrays the caller keeps
        alive; they hold the images after the next synchronise() (after a kfh_compute_frame the renders are carried out by the
        call that closes the frame - the next compute_frame - so that its preprocessing still rides on this frame's ICP launches)."""
        u8 = C.POINTER(C.c_uint8)
        for a in (depth_out, track_out, volume_out):
            assert a.dtype == np.uint8 and a.flags.c_contiguous and a.shape == (self.H, self.W, 4)
        self.L.kfh_render_depth(self.h, depth_out.ctypes.data_as(u8), self.W, self.H)
        self.L.kfh_render_track(self.h, track_out.ctypes.data_as(u8), self.W, self.H)
        self.L.kfh_render_volume(self.h, volume_out.ctypes.data_as(u8), self.W, self.H, frame, rate, _capi.f32(k), largestep)

    def dump_volume(self, filename):
        return self.L.kfh_dump_volume(self.h, filename.encode() if filename else None)

    # ---- accessors (kernels.h:141-194)
    def get_pose(self):
        m = np.zeros((4, 4), np.float32)
        self.L.kfh_get_pose(self.h, m.ctypes.data_as(C.POINTER(C.c_float)))
        return m

    def set_pose(self, m):
        m = np.ascontiguousarray(m, np.float32)
        self.L.kfh_set_pose(self.h, m.ctypes.data_as(C.POINTER(C.c_float)))

    def get_position(self):
        p = np.zeros(3, np.float32)
        self.L.kfh_get_position(self.h, p.ctypes.data_as(C.POINTER(C.c_float)))
        return p

    def get_tracked(self):
        return self._verdict(self.L.kfh_get_tracked(self.h), "kfh_get_tracked")

    def get_integrated(self):
        return self._verdict(self.L.kfh_get_integrated(self.h), "kfh_get_integrated")

    def set_view_pose(self, m=None):
        if m is None:
            self.L.kfh_set_view_pose(self.h, None)
        else:
            m = np.ascontiguousarray(m, np.float32)
            self.L.kfh_set_view_pose(self.h, m.ctypes.data_as(C.POINTER(C.c_float)))

    def get_model_dimensions(self):
        return self.volume_dimensions

    def get_model_resolution(self):
        return self.volume_resolution

    def get_computation_resolution(self):
        return (self.W, self.H)

    def icp_iterations(self):
        return int(self.L.kfh_icp_iterations(self.h))

    def icp_launches_per_frame(self):
        """launches of a tracked frame's ICP loop as this context queues it"""
        return int(self.L.kfh_icp_launches_per_frame(self.h))

    def icp_level_iterations(self):
        """executed ICP iterations by pyramid level (index 0 = full resolution)"""
        out = (C.c_uint64 * 8)()   # KFH_MAX_LEVELS
        self.L.kfh_icp_level_iterations(self.h, out)
        return [int(v) for v in out][:len(self.pyramid)]

    def solve_fallbacks(self):
        return int(self.L.kfh_solve_fallbacks(self.h))

    def barrier_timeouts(self):
        return int(self.L.kfh_barrier_timeouts(self.h))

    def integrate_updated(self):
        return int(self.L.kfh_integrate_updated(self.h))

    PHASES = ("preprocess", "pyramid", "icp", "integrate", "raycast", "render_depth", "render_track", "render_volume")

    def set_profiling(self, phases=(), every=1, rotate=False):
        """Time the named phases with HIP events; `every` > 1 samples every n-th queued frame (an event record stalls
        the queue for a few microseconds, see kfusion_hip.h); `rotate` times one phase per sampled frame."""
        mask = 0
        for p in phases:
            mask |= 1 << self.PHASES.index(p)
        self.L.kfh_set_profiling(self.h, mask)
        self.L.kfh_set_profiling_stride(self.h, max(1, int(every)))
        self.L.kfh_set_profiling_rotation(self.h, 1 if rotate else 0)

    def read_profile(self):
        """{phase: (total_ms, launches)} measured with HIP events on the backend's stream."""
        ms = (C.c_double * len(self.PHASES))()
        n = (C.c_uint64 * len(self.PHASES))()
        self.L.kfh_read_profile(self.h, ms, n)
        return {p: (float(ms[i]), int(n[i])) for i, p in enumerate(self.PHASES)}

    def frame_log(self, n):
        poses = np.zeros((n, 4, 4), np.float32)
        tracked = np.zeros(n, np.int32)
        integrated = np.zeros(n, np.int32)
        r = self.L.kfh_read_frame_log(self.h, n, poses.ctypes.data_as(C.POINTER(C.c_float)),
                                      tracked.ctypes.data_as(C.POINTER(C.c_int32)),
                                      integrated.ctypes.data_as(C.POINTER(C.c_int32)))
        if r < 0:
            raise RuntimeError("kfh_read_frame_log failed")
        return poses, tracked, integrated

    # ---- buffers (parity checks)
    def buffer(self, name):
        n = self.L.kfh_read_buffer(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        raw = np.zeros(n, np.uint8)
        self.L.kfh_read_buffer(self.h, name.encode(), raw.ctypes.data, n)
        lvl = int(name[-1]) if name[-1].isdigit() else 0
        w, h = self.W >> lvl, self.H >> lvl
        if name == "volume":
            r = self.volume_resolution
            return raw.view(np.int16).reshape(r[2], r[1], r[0], 2)
        if name == "trackResult":
            return raw.view(np.int32).reshape(self.H, self.W)
        if name in ("pose", "oldPose", "raycastPose"):
            return raw.view(np.float32).reshape(4, 4)
        if name == "reduction":
            return raw.view(np.float32)
        if name.startswith("scaledDepth") or name == "floatDepth":
            return raw.view(np.float32).reshape(h, w)
        return raw.view(np.float32).reshape(h, w, 4)

    def write_buffer(self, name, array):
        a = np.ascontiguousarray(array)
        r = self.L.kfh_write_buffer(self.h, name.encode(), a.ctypes.data, a.nbytes)
        if r < 0:
            raise ValueError("kfh_write_buffer(%s) failed" % name)

    def device_pointer(self, name):
        return self.L.kfh_device_pointer(self.h, name.encode())
