"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liborc*.so.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The
shipped package (slambench_amd/) never imports this module.  PARITY UNPINNED: see
oracle/orc_math.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

TRACK_DTYPE = np.dtype([("result", "<i4"), ("error", "<f4"), ("J", "<f4", (6,))])
assert TRACK_DTYPE.itemsize == 32


class Counters(C.Structure):
    _fields_ = [("integrate_updated", C.c_ulonglong), ("raycast_distinct_voxels", C.c_ulonglong),
                ("icp_iterations", C.c_ulonglong)]


def build(force: bool = False) -> None:
    """Compile the restatement (and oracle/_ref when /root/reference is present)."""
    import sys
    # make decides what is stale; its chatter goes to stderr (bench.py prints ONE line on stdout)
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), stdout=sys.stderr)


_libs = {}


def lib(omp: bool = False):
    name = "liborc_omp.so" if omp else "liborc.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    fp, u16p, sp, u8p, vp = (C.POINTER(C.c_float), C.POINTER(C.c_ushort), C.POINTER(C.c_short), C.POINTER(C.c_ubyte), C.c_void_p)
    u, i, f = C.c_uint, C.c_int, C.c_float
    u3, f3 = C.POINTER(C.c_uint), C.POINTER(C.c_float)
    cp = C.POINTER(Counters)
    sig = {
        "orc_gaussian": (None, [fp, i, f]),
        "orc_init_volume": (None, [sp, u, u, u]),
        "orc_mm2meters": (i, [fp, u, u, u16p, u, u]),
        "orc_bilateral_filter": (None, [fp, fp, u, u, fp, f, i]),
        "orc_half_sample": (None, [fp, fp, u, u, f, i]),
        "orc_depth2vertex": (None, [fp, fp, u, u, fp]),
        "orc_vertex2normal": (None, [fp, fp, u, u]),
        "orc_track": (None, [vp, fp, fp, u, u, fp, fp, u, u, fp, fp, f, f, i]),
        "orc_reduce": (None, [fp, vp, u, u, u, u]),
        "orc_update_pose": (i, [fp, fp, f]),
        "orc_check_pose": (i, [fp, fp, fp, u, u, f]),
        "orc_solve6": (None, [fp, C.POINTER(C.c_double)]),
        "orc_se3_exp": (None, [C.POINTER(C.c_double), fp]),
        "orc_h_fminf": (f, [f, f]),
        "orc_h_fmaxf": (f, [f, f]),
        "orc_h_clampf": (f, [f, f, f]),
        "orc_h_border_clamp": (i, [i, i, i]),
        "orc_h_clamp_int": (i, [i, i, i]),
        "orc_h_dot3": (f, [fp, fp]),
        "orc_h_cross3": (None, [fp, fp, fp]),
        "orc_h_length3": (f, [fp]),
        "orc_h_normalize3": (None, [fp, fp]),
        "orc_h_floor_frac": (None, [f, fp, fp, C.POINTER(C.c_int)]),
        "orc_h_constants": (None, [fp]),
        "orc_inverse": (None, [fp, fp]),
        "orc_matmul": (None, [fp, fp, fp]),
        "orc_integrate": (None, [sp, u3, f3, fp, u, u, fp, fp, f, f, cp]),
        "orc_raycast": (None, [fp, fp, u, u, sp, u3, f3, fp, f, f, f, f, cp]),
        "orc_render_depth": (None, [u8p, fp, u, u, f, f]),
        "orc_render_track": (None, [u8p, vp, u, u]),
        "orc_render_volume": (None, [u8p, u, u, sp, u3, f3, fp, f, f, f, f, f3, f3]),
        "orc_kfusion_create": (vp, [u, u, u3, f3, f3, C.POINTER(C.c_int), i]),
        "orc_kfusion_destroy": (None, [vp]),
        "orc_kfusion_reset": (None, [vp]),
        "orc_kfusion_preprocessing": (i, [vp, u16p, u, u]),
        "orc_kfusion_tracking": (i, [vp, fp, f, u, u]),
        "orc_kfusion_integration": (i, [vp, fp, u, f, u]),
        "orc_kfusion_raycasting": (i, [vp, fp, f, u]),
        "orc_kfusion_render_depth": (None, [vp, u8p, u, u]),
        "orc_kfusion_render_track": (None, [vp, u8p, u, u]),
        "orc_kfusion_render_volume": (None, [vp, u8p, u, u, i, i, fp, f]),
        "orc_kfusion_get_pose": (None, [vp, fp]),
        "orc_kfusion_set_pose": (None, [vp, fp]),
        "orc_kfusion_get_position": (None, [vp, fp]),
        "orc_kfusion_dump_volume": (i, [vp, C.c_char_p]),
        "orc_kfusion_counters": (cp, [vp]),
        "orc_kfusion_set_counting": (None, [vp, i]),
        "orc_set_omp_reduce_mode": (None, [i]),
        "orc_set_omp_threads": (None, [i]),
        "orc_set_reduce_emulation": (None, [i]),
        "orc_kfusion_buffer": (vp, [vp, C.c_char_p, C.POINTER(C.c_ulonglong)]),
    }
    for name_, (res, args) in sig.items():
        fn = getattr(L, name_)
        fn.restype = res
        fn.argtypes = args
    _libs[name] = L
    return L


def set_omp_threads(n: int) -> None:
    """Thread count of the -fopenmp build from now on."""
    lib(True).orc_set_omp_threads(int(n))


def set_omp_reduce_mode(reference_style: bool) -> None:
    """-fopenmp build: True = the reduction is parallelised like the reference's OpenMP build (re-associated);
    False (default) = bit-identical to the serial build."""
    lib(True).orc_set_omp_reduce_mode(1 if reference_style else 0)


def set_reduce_emulation(threads: int, omp: bool = True) -> None:
    """> 0: reduce like the reference's OpenMP build (cpp:295-296) with that many threads under a FIXED schedule and combine
    order (deterministic, whatever the host has); 0 = off.  Per library (serial / -fopenmp build)."""
    lib(omp).orc_set_reduce_emulation(int(threads))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def fptr(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return _p(a, C.c_float)


def arr3(v, t):
    return (t * 3)(*v)


# ---------------------------------------------------------------- kernel-level wrappers
def gaussian(radius=2, delta=4.0, omp=False):
    g = np.zeros(2 * radius + 1, np.float32)
    lib(omp).orc_gaussian(fptr(g), radius, delta)
    return g


def mm2meters(depth_mm, out_w, out_h, omp=False):
    depth_mm = np.ascontiguousarray(depth_mm, np.uint16)
    h, w = depth_mm.shape
    out = np.zeros((out_h, out_w), np.float32)
    rc = lib(omp).orc_mm2meters(fptr(out), out_w, out_h, _p(depth_mm, C.c_ushort), w, h)
    if rc:
        raise ValueError("Invalid ratio.")
    return out


def bilateral_filter(depth, g=None, e_d=0.1, r=2, omp=False):
    depth = np.ascontiguousarray(depth, np.float32)
    g = gaussian() if g is None else np.ascontiguousarray(g, np.float32)
    out = np.zeros_like(depth)
    lib(omp).orc_bilateral_filter(fptr(out), fptr(depth), depth.shape[1], depth.shape[0], fptr(g), e_d, r)
    return out


def half_sample(depth, e_d=np.float32(0.1) * 3, r=1, omp=False):
    depth = np.ascontiguousarray(depth, np.float32)
    h, w = depth.shape
    out = np.zeros((h // 2, w // 2), np.float32)
    lib(omp).orc_half_sample(fptr(out), fptr(depth), w, h, float(e_d), r)
    return out


def inverse_camera_matrix(k):
    k = np.asarray(k, np.float32)
    m = np.zeros((4, 4), np.float32)
    m[0, 0] = np.float32(1.0) / k[0]; m[0, 2] = -k[2] / k[0]
    m[1, 1] = np.float32(1.0) / k[1]; m[1, 2] = -k[3] / k[1]
    m[2, 2] = 1; m[3, 3] = 1
    return m


def camera_matrix(k):
    k = np.asarray(k, np.float32)
    m = np.zeros((4, 4), np.float32)
    m[0, 0] = k[0]; m[0, 2] = k[2]; m[1, 1] = k[1]; m[1, 2] = k[3]; m[2, 2] = 1; m[3, 3] = 1
    return m


def depth2vertex(depth, invK, omp=False):
    depth = np.ascontiguousarray(depth, np.float32)
    invK = np.ascontiguousarray(invK, np.float32)
    out = np.zeros(depth.shape + (3,), np.float32)
    lib(omp).orc_depth2vertex(fptr(out), fptr(depth), depth.shape[1], depth.shape[0], fptr(invK))
    return out


def vertex2normal(vertex, out=None, omp=False):
    vertex = np.ascontiguousarray(vertex, np.float32)
    out = np.zeros_like(vertex) if out is None else out
    lib(omp).orc_vertex2normal(fptr(out), fptr(vertex), vertex.shape[1], vertex.shape[0])
    return out


def track(in_vertex, in_normal, ref_vertex, ref_normal, pose, view, out=None, dist_threshold=0.1,
          normal_threshold=0.8, have_reference=True, omp=False):
    in_vertex = np.ascontiguousarray(in_vertex, np.float32); in_normal = np.ascontiguousarray(in_normal, np.float32)
    ref_vertex = np.ascontiguousarray(ref_vertex, np.float32); ref_normal = np.ascontiguousarray(ref_normal, np.float32)
    pose = np.ascontiguousarray(pose, np.float32); view = np.ascontiguousarray(view, np.float32)
    ih, iw = in_vertex.shape[:2]
    rh, rw = ref_vertex.shape[:2]
    if out is None:
        out = np.zeros((rh, rw), TRACK_DTYPE)
    lib(omp).orc_track(out.ctypes.data, fptr(in_vertex), fptr(in_normal), iw, ih, fptr(ref_vertex), fptr(ref_normal), rw, rh,
                       fptr(pose), fptr(view), dist_threshold, normal_threshold, 1 if have_reference else 0)
    return out


def reduce(track_data, w, h, omp=False):
    jh, jw = track_data.shape
    out = np.zeros((8, 32), np.float32)
    lib(omp).orc_reduce(fptr(out), track_data.ctypes.data, jw, jh, w, h)
    return out


def update_pose(pose, reduction, icp_threshold=1e-5, omp=False):
    pose = np.array(pose, np.float32, copy=True)
    reduction = np.ascontiguousarray(reduction, np.float32)
    ok = lib(omp).orc_update_pose(fptr(pose), fptr(reduction), icp_threshold)
    return pose, bool(ok)


def check_pose(pose, old_pose, reduction, w, h, track_threshold=0.15, omp=False):
    pose = np.array(pose, np.float32, copy=True)
    old_pose = np.ascontiguousarray(old_pose, np.float32); reduction = np.ascontiguousarray(reduction, np.float32)
    ok = lib(omp).orc_check_pose(fptr(pose), fptr(old_pose), fptr(reduction), w, h, track_threshold)
    return pose, bool(ok)


def solve6(vals27, omp=False):
    vals27 = np.ascontiguousarray(vals27, np.float32)
    x = np.zeros(6, np.float64)
    lib(omp).orc_solve6(fptr(vals27), _p(x, C.c_double))
    return x


def se3_exp(mu6, omp=False):
    mu6 = np.ascontiguousarray(mu6, np.float64)
    out = np.zeros((4, 4), np.float32)
    lib(omp).orc_se3_exp(_p(mu6, C.c_double), fptr(out))
    return out


def inverse(m, omp=False):
    m = np.ascontiguousarray(m, np.float32)
    out = np.zeros((4, 4), np.float32)
    with np.errstate(all="ignore"):
        lib(omp).orc_inverse(fptr(m), fptr(out))
    return out


def matmul(a, b, omp=False):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros((4, 4), np.float32)
    lib(omp).orc_matmul(fptr(a), fptr(b), fptr(out))
    return out


def init_volume(res, omp=False):
    vol = np.zeros((res[2], res[1], res[0], 2), np.int16)
    lib(omp).orc_init_volume(_p(vol, C.c_short), res[0], res[1], res[2])
    return vol


def integrate(vol, dim, depth, inv_track, K, mu=0.1, maxweight=100.0, omp=False):
    """In place on vol (z, y, x, 2) int16.  Returns the number of updated voxels."""
    assert vol.dtype == np.int16 and vol.flags.c_contiguous
    depth = np.ascontiguousarray(depth, np.float32)
    inv_track = np.ascontiguousarray(inv_track, np.float32); K = np.ascontiguousarray(K, np.float32)
    cnt = Counters()
    size = arr3((vol.shape[2], vol.shape[1], vol.shape[0]), C.c_uint)
    lib(omp).orc_integrate(_p(vol, C.c_short), size, arr3(dim, C.c_float), fptr(depth), depth.shape[1], depth.shape[0],
                           fptr(inv_track), fptr(K), mu, maxweight, C.byref(cnt))
    return int(cnt.integrate_updated)


def raycast(vol, dim, view, w, h, near=0.4, far=4.0, step=None, largestep=0.075, vertex=None, normal=None,
            count=False, omp=False):
    assert vol.dtype == np.int16 and vol.flags.c_contiguous
    view = np.ascontiguousarray(view, np.float32)
    vertex = np.zeros((h, w, 3), np.float32) if vertex is None else vertex
    normal = np.zeros((h, w, 3), np.float32) if normal is None else normal
    res = (vol.shape[2], vol.shape[1], vol.shape[0])
    if step is None:
        step = np.float32(min(dim)) / np.float32(max(res))
    cnt = Counters()
    lib(omp).orc_raycast(fptr(vertex), fptr(normal), w, h, _p(vol, C.c_short), arr3(res, C.c_uint), arr3(dim, C.c_float),
                         fptr(view), near, far, float(step), largestep, C.byref(cnt) if count else None)
    return (vertex, normal, int(cnt.raycast_distinct_voxels)) if count else (vertex, normal)


def render_depth(depth, near=0.4, far=4.0, omp=False):
    depth = np.ascontiguousarray(depth, np.float32)
    out = np.zeros(depth.shape + (4,), np.uint8)
    lib(omp).orc_render_depth(_p(out, C.c_ubyte), fptr(depth), depth.shape[1], depth.shape[0], near, far)
    return out


def render_track(track_data, omp=False):
    out = np.zeros(track_data.shape + (4,), np.uint8)
    lib(omp).orc_render_track(_p(out, C.c_ubyte), track_data.ctypes.data, track_data.shape[1], track_data.shape[0])
    return out


def render_volume(vol, dim, view, w, h, near=0.4, far=8.0, step=None, largestep=0.075, light=(1, 1, -1),
                  ambient=(0.1, 0.1, 0.1), omp=False):
    view = np.ascontiguousarray(view, np.float32)
    res = (vol.shape[2], vol.shape[1], vol.shape[0])
    if step is None:
        step = np.float32(min(dim)) / np.float32(max(res))
    out = np.zeros((h, w, 4), np.uint8)
    lib(omp).orc_render_volume(_p(out, C.c_ubyte), w, h, _p(vol, C.c_short), arr3(res, C.c_uint), arr3(dim, C.c_float),
                               fptr(view), near, far, float(step), largestep, arr3(light, C.c_float), arr3(ambient, C.c_float))
    return out


# ---------------------------------------------------------------- pipeline object
class OracleKfusion:
    """CPU twin of the reference's class Kfusion (kernels.h:83-195)."""

    def __init__(self, size, vol_res, vol_dim, init_pose, pyramid=(10, 5, 4), omp=False):
        self.L = lib(omp)
        self.W, self.H = int(size[0]), int(size[1])
        self.vol_res = tuple(int(v) for v in vol_res)
        self.vol_dim = tuple(float(v) for v in vol_dim)
        self.pyramid = list(pyramid)
        pyr = (C.c_int * len(pyramid))(*pyramid)
        self.h = self.L.orc_kfusion_create(self.W, self.H, arr3(self.vol_res, C.c_uint), arr3(self.vol_dim, C.c_float),
                                           arr3(init_pose, C.c_float), pyr, len(pyramid))

    def close(self):
        if self.h:
            self.L.orc_kfusion_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _k(k):
        return (C.c_float * 4)(*k)

    def reset(self):
        """Kfusion::reset (cpp:135-137): the volume back to its initial state, nothing else"""
        self.L.orc_kfusion_reset(self.h)

    def preprocessing(self, depth_mm):
        depth_mm = np.ascontiguousarray(depth_mm, np.uint16)
        return self.L.orc_kfusion_preprocessing(self.h, _p(depth_mm, C.c_ushort), depth_mm.shape[1], depth_mm.shape[0]) == 1

    def tracking(self, k, icp_threshold, tracking_rate, frame):
        return bool(self.L.orc_kfusion_tracking(self.h, self._k(k), icp_threshold, tracking_rate, frame))

    def integration(self, k, integration_rate, mu, frame):
        return bool(self.L.orc_kfusion_integration(self.h, self._k(k), integration_rate, mu, frame))

    def raycasting(self, k, mu, frame):
        return bool(self.L.orc_kfusion_raycasting(self.h, self._k(k), mu, frame))

    def render_depth(self):
        out = np.zeros((self.H, self.W, 4), np.uint8)
        self.L.orc_kfusion_render_depth(self.h, _p(out, C.c_ubyte), self.W, self.H)
        return out

    def render_track(self):
        out = np.zeros((self.H, self.W, 4), np.uint8)
        self.L.orc_kfusion_render_track(self.h, _p(out, C.c_ubyte), self.W, self.H)
        return out

    def render_volume(self, frame, rate, k, largestep, out=None):
        out = np.zeros((self.H, self.W, 4), np.uint8) if out is None else out
        self.L.orc_kfusion_render_volume(self.h, _p(out, C.c_ubyte), self.W, self.H, frame, rate, self._k(k), largestep)
        return out

    def get_pose(self):
        m = np.zeros((4, 4), np.float32)
        self.L.orc_kfusion_get_pose(self.h, fptr(m))
        return m

    def set_pose(self, m):
        m = np.ascontiguousarray(m, np.float32)
        self.L.orc_kfusion_set_pose(self.h, fptr(m))

    def get_position(self):
        p = np.zeros(3, np.float32)
        self.L.orc_kfusion_get_position(self.h, fptr(p))
        return p

    def dump_volume(self, path):
        return self.L.orc_kfusion_dump_volume(self.h, path.encode())

    def set_counting(self, raycast_distinct=True):
        self.L.orc_kfusion_set_counting(self.h, 1 if raycast_distinct else 0)

    def counters(self):
        c = self.L.orc_kfusion_counters(self.h).contents
        return {"integrate_updated": int(c.integrate_updated), "raycast_distinct_voxels": int(c.raycast_distinct_voxels),
                "icp_iterations": int(c.icp_iterations)}

    def buffer(self, name):
        """Copy of an internal buffer as a numpy array (shape by name)."""
        n = C.c_ulonglong(0)
        p = self.L.orc_kfusion_buffer(self.h, name.encode(), C.byref(n))
        if not p or n.value == 0:
            raise KeyError(name)
        raw = np.frombuffer(C.string_at(p, n.value), dtype=np.uint8).copy()
        if name == "trackingResult":
            return raw.view(TRACK_DTYPE).reshape(self.H, self.W)
        if name == "volume":
            return raw.view(np.int16).reshape(self.vol_res[2], self.vol_res[1], self.vol_res[0], 2)
        if name in ("raycastPose", "oldPose"):
            return raw.view(np.float32).reshape(4, 4)
        if name == "reduction":
            return raw.view(np.float32).reshape(8, 32)
        f = raw.view(np.float32)
        lvl = int(name[-1]) if name[-1].isdigit() else 0
        w, h = self.W >> lvl, self.H >> lvl
        if name.startswith("scaledDepth") or name == "floatDepth":
            return f[: w * h].reshape(h, w)
        return f[: w * h * 3].reshape(h, w, 3)
