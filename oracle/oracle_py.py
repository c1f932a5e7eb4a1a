"""ctypes view of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by raytracing_amd."""
import ctypes as C
import os
from pathlib import Path

import numpy as np

from raytracing_amd._ffi import CameraBuilderStruct
from raytracing_amd.api import CameraSettings, SceneBuilder

_PATH = Path(__file__).resolve().parent / "liboracle.so"
MODE_REF_EXACT, MODE_BOOK3 = 0, 1
RNG_PHILOX24, RNG_PHILOX53, RNG_MT64 = 0, 1, 2
NCOUNTERS = 10
COUNTER_NAMES = ("segments", "box_tests", "sphere_tests", "sphere_accepts", "quad_plane_tests",
                 "quad_interior_tests", "dropped", "camera_rays", "zero_weight_segments", "reserved")


class RenderParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("rng_kind", C.c_int32), ("seed", C.c_uint64), ("stratum_begin", C.c_uint32),
                ("stratum_end", C.c_uint32), ("sanitize", C.c_int32), ("threads", C.c_int32), ("x0", C.c_uint32),
                ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


def load():
    if not _PATH.exists():
        raise ImportError(f"{_PATH} missing: run `make -C oracle`")
    lib = C.CDLL(os.fspath(_PATH))
    lib.orc_scene_new.restype = C.c_void_p
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_last_error.argtypes = [C.c_void_p]
    lib.orc_schlick.restype = C.c_double
    lib.orc_schlick.argtypes = [C.c_double, C.c_double]
    lib.orc_interval_clamp.restype = C.c_double
    lib.orc_interval_clamp.argtypes = [C.c_double] * 3
    for n in ("orc_interval_contains", "orc_interval_surrounds"):
        getattr(lib, n).argtypes = [C.c_double] * 3
    lib.orc_interval_overlap.argtypes = [C.c_double] * 4 + [C.POINTER(C.c_double)]
    lib.orc_interval_expand.argtypes = [C.c_double] * 3 + [C.POINTER(C.c_double)]
    return lib


lib = load()


def scene() -> SceneBuilder:
    return SceneBuilder(lib, "orc_")


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def render(sb: SceneBuilder, cam: CameraSettings, mode=MODE_REF_EXACT, rng=RNG_PHILOX24, seed=1, stratum_begin=0,
           stratum_end=0, sanitize=True, threads=1, window=None, want_sumsq=False):
    """Returns (film_sum (H,W,3) f64, film_sumsq or None, counters dict)."""
    cs = cam.to_struct()
    p = RenderParams(mode, rng, seed, stratum_begin, stratum_end, 1 if sanitize else 0, threads, 0, 0, 0, 0)
    if window:
        p.x0, p.y0, p.x1, p.y1 = window
    H, W = cam.image_height, cam.image_width
    film = np.zeros((H, W, 3), np.float64)
    sq = np.zeros((H, W, 3), np.float64) if want_sumsq else None
    cnt = (C.c_uint64 * NCOUNTERS)()
    rc = lib.orc_render(sb.handle, C.byref(cs), C.byref(p), _dp(film), _dp(sq) if want_sumsq else None, cnt)
    if rc < 0:
        raise RuntimeError(lib.orc_last_error(sb.handle).decode())
    return film, sq, dict(zip(COUNTER_NAMES, list(cnt)))


def render_flagged(sb: SceneBuilder, cam: CameraSettings, radius_below, mode=MODE_BOOK3, rng=RNG_PHILOX24, seed=1, threads=1):
    """render() plus, per pixel, how many of its samples scattered off a sphere smaller than
    `radius_below`: (film_sum (H,W,3) f64, counts (H,W) uint32)."""
    cs = cam.to_struct()
    p = RenderParams(mode, rng, seed, 0, 0, 1, threads, 0, 0, 0, 0)
    H, W = cam.image_height, cam.image_width
    film = np.zeros((H, W, 3), np.float64)
    flagged = np.zeros((H, W), np.uint32)
    lib.orc_render_flagged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_uint32)]
    rc = lib.orc_render_flagged(sb.handle, C.byref(cs), C.byref(p), _dp(film), C.c_double(radius_below),
                                flagged.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc < 0:
        raise RuntimeError(lib.orc_last_error(sb.handle).decode())
    return film, flagged


def camera_rays(sb, cam: CameraSettings, pixel, stratum, seed=1, rng=RNG_PHILOX24):
    cs = cam.to_struct()
    p = RenderParams(0, rng, seed, 0, 0, 1, 1, 0, 0, 0, 0)
    pixel = np.ascontiguousarray(pixel, np.uint32)
    stratum = np.ascontiguousarray(stratum, np.uint32)
    n = len(pixel)
    o = np.empty((n, 3)), np.empty((n, 3)), np.empty(n)
    up = C.POINTER(C.c_uint32)
    rc = lib.orc_camera_rays(sb.handle, C.byref(cs), C.byref(p), n, pixel.ctypes.data_as(up), stratum.ctypes.data_as(up),
                             _dp(o[0]), _dp(o[1]), _dp(o[2]))
    if rc < 0:
        raise RuntimeError(lib.orc_last_error(sb.handle).decode())
    return o


def intersect_batch(sb, o, d, time=None, tmin=0.001, tmax=float("inf"), want_normal=False):
    o = np.ascontiguousarray(o, np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float64).reshape(-1, 3)
    n = len(o)
    tm = np.ascontiguousarray(time, np.float64) if time is not None else None
    hit, prim = np.empty(n, np.int32), np.empty(n, np.int32)
    t, u, v, t2 = np.empty(n), np.empty(n), np.empty(n), np.empty(n)
    nrm = np.empty((n, 3)) if want_normal else None
    ip = C.POINTER(C.c_int32)
    rc = lib.orc_intersect_batch(sb.handle, n, _dp(o), _dp(d), _dp(tm) if tm is not None else None, C.c_double(tmin),
                                 C.c_double(tmax), hit.ctypes.data_as(ip), prim.ctypes.data_as(ip), _dp(t), _dp(u),
                                 _dp(v), _dp(t2), _dp(nrm) if want_normal else None)
    if rc < 0:
        raise RuntimeError(lib.orc_last_error(sb.handle).decode())
    if want_normal:
        return hit, prim, t, u, v, t2, nrm
    return hit, prim, t, u, v, t2


def tex_value(sb, tex, u, v, p):
    out = (C.c_double * 3)()
    lib.orc_tex_value(sb.handle, C.c_int(tex), C.c_double(u), C.c_double(v), (C.c_double * 3)(*p), out)
    return np.array(out[:])


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib.orc_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def philox2x32(ctr, key):
    out = (C.c_uint32 * 2)()
    lib.orc_philox2x32((C.c_uint32 * 2)(*ctr), C.c_uint32(key), out)
    return list(out)
