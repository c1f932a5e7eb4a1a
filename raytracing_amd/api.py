"""Python view of the host scene layer and the device backend.

`SceneBuilder` drives a scene-construction C API that follows the reference's object model
(src/hittable.rs, src/material.rs, src/texture.rs): the product's `rth_*` functions by
default.  It is generic over (library, prefix) because any library exporting the same
construction calls can be driven from one scene description.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import (CameraBuilderStruct, RtCamera, RtRenderParams, RtSceneDesc, RtStats, RT_FLAG_TIME_KERNELS, RT_FLAG_WAVEFRONT, RT_FLAG_REGROUP, RT_FLAG_NO_REGROUP, RT_FLAG_GENERAL_KERNELS,
                   RT_MODE_BOOK3, RT_MODE_REF_EXACT)

SHAPE_PARALLELOGRAM, SHAPE_TRIANGLE, SHAPE_DISC = 0, 1, 2


class RtError(RuntimeError):
    pass


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class SceneBuilder:
    """Builds a scene through `<prefix>tex_*`, `<prefix>mat_*`, `<prefix>sphere`, ... calls."""

    def __init__(self, lib=None, prefix="rth_"):
        self.lib = lib if lib is not None else _ffi.lib
        self.prefix = prefix
        new = self._fn("scene_new")
        new.restype = C.c_void_p
        self.handle = C.c_void_p(new())
        self.world = None
        self.lights = None
        self._solid_cache = {}

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def close(self):
        if self.handle:
            f = self._fn("scene_free")
            f.argtypes = [C.c_void_p]
            f(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            f = self._fn("last_error")
            f.restype = C.c_char_p
            f.argtypes = [C.c_void_p]
            raise RtError(f"{what}: {f(self.handle).decode()}")
        return rc

    def _call(self, name, *args):
        f = self._fn(name)
        f.restype = C.c_int
        return self._check(f(self.handle, *args), name)

    # textures (texture.rs)
    def tex_solid(self, r, g, b):
        return self._call("tex_solid", C.c_double(r), C.c_double(g), C.c_double(b))

    def tex_checker(self, scale, even, odd):
        return self._call("tex_checker", C.c_double(scale), C.c_int(even), C.c_int(odd))

    def tex_image(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, c = rgb.shape
        assert c == 3
        return self._call("tex_image", C.c_uint32(w), C.c_uint32(h), rgb.ctypes.data_as(C.POINTER(C.c_uint8)))

    def tex_noise(self, scale, ranvec, perm):
        ranvec = np.ascontiguousarray(ranvec, dtype=np.float64).reshape(256, 3)
        perm = np.ascontiguousarray(perm, dtype=np.int32).reshape(3, 256)
        return self._call("tex_noise", C.c_double(scale), ranvec.ctypes.data_as(C.POINTER(C.c_double)),
                          perm.ctypes.data_as(C.POINTER(C.c_int32)))

    # materials (material.rs)
    def _tex_of(self, t):
        if isinstance(t, (tuple, list)):
            key = tuple(float(x) for x in t)
            if key not in self._solid_cache:
                self._solid_cache[key] = self.tex_solid(*key)
            return self._solid_cache[key]
        return int(t)

    def lambertian(self, tex_or_color):
        return self._call("mat_lambertian", C.c_int(self._tex_of(tex_or_color)))

    def light(self, tex_or_color):
        return self._call("mat_light", C.c_int(self._tex_of(tex_or_color)))

    def metal(self, color, fuzz):
        return self._call("mat_metal", C.c_double(color[0]), C.c_double(color[1]), C.c_double(color[2]), C.c_double(fuzz))

    def dielectric(self, ior):
        return self._call("mat_dielectric", C.c_double(ior))

    def isotropic(self, tex_or_color):
        return self._call("mat_isotropic", C.c_int(self._tex_of(tex_or_color)))

    # hittables (hittable.rs, boundingbox.rs)
    def sphere(self, center, radius, mat, center_vec=(0.0, 0.0, 0.0)):
        return self._call("sphere", _d3(center), _d3(center_vec), C.c_double(radius), C.c_int(mat))

    def quad(self, corner, u, v, mat, shape=SHAPE_PARALLELOGRAM):
        return self._call("quad", C.c_int(shape), _d3(corner), _d3(u), _d3(v), C.c_int(mat))

    def box3(self, a, b, mat):
        return self._call("box3", _d3(a), _d3(b), C.c_int(mat))

    def _ids(self, objs):
        return (C.c_int * len(objs))(*[int(o) for o in objs]), C.c_int(len(objs))

    def list(self, objs: Sequence[int]):
        return self._call("list", *self._ids(objs))

    def bvh(self, objs: Sequence[int]):
        return self._call("bvh", *self._ids(objs))

    def translate(self, obj, offset):
        return self._call("translate", C.c_int(obj), _d3(offset))

    def rotate_y(self, obj, degrees):
        return self._call("rotate_y", C.c_int(obj), C.c_double(math.radians(degrees)))

    def medium(self, boundary, density, tex_or_color):
        return self._call("medium", C.c_int(boundary), C.c_double(density), C.c_int(self._tex_of(tex_or_color)))

    def set_world(self, obj):
        self.world = obj
        return self._call("set_world", C.c_int(obj))

    def set_lights(self, obj):
        self.lights = obj
        return self._call("set_lights", C.c_int(-1 if obj is None else obj))

    def prim_count(self):
        return self._call("prim_count")

    def bvh_stats(self, obj):
        n, d = C.c_int(), C.c_int()
        self._call("bvh_stats", C.c_int(obj), C.byref(n), C.byref(d))
        return n.value, d.value

    # product-only calls (prefix rth_)
    def load_toml(self, text: str):
        """ConfigModel::from_str + as_world (config.rs:628-706): returns the world list's object id."""
        return self._call("load_toml", C.c_char_p(text.encode()))

    def list_add(self, lst, obj):
        return self._call("list_add", C.c_int(lst), C.c_int(obj))

    def material_by_name(self, name):
        return self._call("material_by_name", C.c_char_p(name.encode()))

    def flatten(self):
        out = C.POINTER(RtSceneDesc)()
        self._call("flatten", C.byref(out))
        return out


@dataclass
class CameraSettings:
    """CameraBuilder fields and defaults (camera.rs:183-202)."""
    image_width: int = 400
    image_height: int = 200
    vfov: float = 90.0
    samples_per_px: int = 10
    max_depth: int = 10
    background: Optional[Sequence[float]] = None  # None = Background::Sky
    camera_center: Sequence[float] = (0.0, 0.0, 0.0)
    camera_target: Sequence[float] = (0.0, 0.0, -1.0)
    vup: Sequence[float] = (0.0, 1.0, 0.0)
    defocus_angle: float = 0.0
    focal_length: float = 1.0
    antialiasing: str = "square"  # AntialiasingType (camera.rs:204-211): "square" | "disc"

    def with_aspect_ratio(self, width, aspect_ratio):
        """camera.rs:78-92"""
        self.image_width = width
        self.image_height = int(round(width / aspect_ratio))
        return self

    @property
    def sqrt_spp(self):
        return int(math.floor(math.sqrt(self.samples_per_px)))

    @property
    def effective_spp(self):
        return self.sqrt_spp ** 2

    def to_struct(self):
        b = CameraBuilderStruct()
        b.image_width, b.image_height = self.image_width, self.image_height
        b.vfov_deg = self.vfov
        b.samples_per_px, b.max_depth = self.samples_per_px, self.max_depth
        b.background_sky = 1 if self.background is None else 0
        bg = self.background or (0.0, 0.0, 0.0)
        for i in range(3):
            b.background[i] = bg[i]
            b.center[i] = self.camera_center[i]
            b.target[i] = self.camera_target[i]
            b.vup[i] = self.vup[i]
        b.antialiasing = {"square": 0, "disc": 1}[self.antialiasing]
        b.defocus_angle_deg = self.defocus_angle
        b.focal_length = self.focal_length
        return b

    def build(self) -> RtCamera:
        """CameraBuilder::build -> Camera::build; raises with the builder's error list."""
        cam = RtCamera()
        err = C.create_string_buffer(4096)
        b = self.to_struct()
        if _ffi.lib.rth_camera_build(C.byref(b), C.byref(cam), err, 4096) < 0:
            raise RtError(err.value.decode())
        return cam


@dataclass
class Stats:
    segments: int = 0
    camera_rays: int = 0
    dropped: int = 0
    iterations: int = 0
    launches: tuple = (0, 0, 0, 0, 0)           # generate, extend, shade, scan, paths
    timed_launches: tuple = (0, 0, 0, 0, 0)
    kernel_ms: tuple = (0.0, 0.0, 0.0, 0.0, 0.0)  # HIP-event time of the timed launches
    render_ms: float = 0.0
    regrouped: bool = False  # the persistent kernel ran in its regrouping variant


class Device:
    """One GPU (rt_handle)."""

    def __init__(self, device_id=0):
        self.lib = _ffi.lib
        h = C.c_void_p()
        if self.lib.rt_create(device_id, C.byref(h)) < 0:
            raise RtError(self.lib.rt_last_error(None).decode())
        self.h = h
        self.device_id = device_id
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise RtError(f"{what}: {self.lib.rt_last_error(self.h).decode()}")

    def set_stream(self, hip_stream_ptr):
        self._check(self.lib.rt_set_stream(self.h, C.c_void_p(hip_stream_ptr)), "rt_set_stream")

    def set_accel(self, builder="auto", width=4, fp32_boxes=None, box_leaves=True, post_tests=True):
        """Structure built by the next upload: builder "reference" | "sah" | "auto", width 2 | 4;
        fp32_boxes: True = 128-byte wide nodes with fp32 child boxes, False = 64-byte nodes with byte
        boxes, None = chosen by the size of the tree; box_leaves=False keeps the six faces of an
        axis-aligned box six quad leaves."""
        b = {"reference": _ffi.RT_BUILD_REFERENCE, "sah": _ffi.RT_BUILD_SAH, "auto": _ffi.RT_BUILD_AUTO}[builder]
        if fp32_boxes is not None:
            b |= _ffi.RT_BUILD_FP32_BOXES if fp32_boxes else _ffi.RT_BUILD_BYTE_BOXES
        if not box_leaves:
            b |= _ffi.RT_BUILD_NO_BOX_LEAVES
        if not post_tests:  # primitives that enclose the scene stay in the tree
            b |= _ffi.RT_BUILD_NO_POST_TESTS
        self._check(self.lib.rt_set_accel(self.h, b, width), "rt_set_accel")

    def accel_info(self):
        a = _ffi.RtAccelInfo()
        self._check(self.lib.rt_get_accel_info(self.h, C.byref(a)), "rt_get_accel_info")
        return dict(kind=("flat", "bvh2", "bvh4")[a.kind], nodes=a.nodes, depth=a.depth, stack_entries=a.stack_entries, cost=a.cost,
                    flat_quads=a.flat_quads, flat_boxes=a.flat_boxes, flat_alone_spheres=a.flat_alone_spheres, node_bytes=a.node_bytes,
                    flat_boxes_aligned=a.flat_boxes_aligned, flat_boxes_upright=a.flat_boxes_upright, flat_rects=a.flat_rects,
                    box_leaves=a.box_leaves, leaves=a.leaves, post_tests=a.post_tests, walk_cost=a.walk_cost)

    def upload(self, scene: SceneBuilder):
        desc = scene.flatten()
        self._check(self.lib.rt_upload_scene(self.h, desc), "rt_upload_scene")
        self._scene = scene  # keeps the flat arrays alive
        return desc.contents

    def upload_desc(self, desc):
        self._check(self.lib.rt_upload_scene(self.h, desc), "rt_upload_scene")

    def film_resize(self, w, h):
        self._check(self.lib.rt_film_resize(self.h, w, h), "rt_film_resize")
        self.width, self.height = w, h

    def film_bind(self, device_ptr, w, h):
        self._check(self.lib.rt_film_bind(self.h, C.c_void_p(device_ptr), w, h), "rt_film_bind")
        self.width, self.height = w, h

    def film_clear(self):
        self._check(self.lib.rt_film_clear(self.h), "rt_film_clear")

    def last_kernel_variant(self):
        """RT_VARIANT_* bits of the persistent kernel the last render took (0 after a wavefront render)."""
        return int(self.lib.rt_last_kernel_variant(self.h))

    def render(self, cam: RtCamera, mode=RT_MODE_REF_EXACT, seed=1, stratum_begin=0, stratum_end=0, pool_size=0,
               time_kernels=False, time_every=0, wavefront=False, refill_min=0, waves_per_simd=0, regroup=None, general_kernels=False):
        """Adds strata [begin, end) of every pixel into the film (camera.rs:386-401).
        wavefront=True takes the ray-queue pipeline instead of the persistent path kernel.
        regroup: True / False forces the persistent kernel's regrouping or lane-bound variant on a
        flat world, None leaves the choice to the library.  general_kernels=True takes the kernels
        compiled for any scene instead of the variants compiled for what this scene lacks."""
        p = RtRenderParams()
        p.mode, p.seed = mode, seed
        p.flags = (RT_FLAG_TIME_KERNELS if time_kernels else 0) | (RT_FLAG_WAVEFRONT if wavefront else 0) | \
            (0 if regroup is None else (RT_FLAG_REGROUP if regroup else RT_FLAG_NO_REGROUP)) | \
            (RT_FLAG_GENERAL_KERNELS if general_kernels else 0)
        p.refill_min, p.waves_per_simd = refill_min, waves_per_simd
        p.stratum_begin, p.stratum_end, p.pool_size, p.time_every = stratum_begin, stratum_end, pool_size, time_every
        self._check(self.lib.rt_render(self.h, C.byref(cam), C.byref(p)), "rt_render")
        self.width, self.height = cam.image_width, cam.image_height
        return self.stats()

    def stats(self) -> Stats:
        s = RtStats()
        self._check(self.lib.rt_get_stats(self.h, C.byref(s)), "rt_get_stats")
        return Stats(s.segments, s.camera_rays, s.dropped, s.iterations, tuple(s.launches), tuple(s.timed_launches),
                     tuple(s.kernel_ms), s.render_ms, bool(s.regrouped))

    def film(self, scale=1.0):
        """(H, W, 3) float32 = film sums * scale."""
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._check(self.lib.rt_film_download(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(scale)),
                    "rt_film_download")
        return out

    def film_rgb8(self, scale, gamma=False, no_alloc=False, out=None):
        """(H, W, 3) uint8: the reference's 8-bit rule applied to sums * scale on the device
        (gamma=True is what PpmWriter writes, False what PngWriter writes).  no_alloc=True takes
        the entry point that stages through the library's fixed device array (films up to
        3840 x 2160) and makes no allocating call; `out` receives the bytes if given."""
        if out is None:
            out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        if not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous and out.flags.writeable
                and out.size == self.height * self.width * 3):
            raise ValueError(f"out must be a writable C-contiguous uint8 array of {self.height * self.width * 3} bytes")
        fn, name = (self.lib.rt_film_quantize_noalloc, "rt_film_quantize_noalloc") if no_alloc else \
            (self.lib.rt_film_quantize, "rt_film_quantize")
        self._check(fn(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_double(scale), 1 if gamma else 0), name)
        return out

    def film_upload(self, sums):
        """Restores film sums: (H, W, 3) float32 as returned by film(1.0)."""
        a = np.ascontiguousarray(sums, dtype=np.float32)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("film sums must have shape (H, W, 3)")
        self._check(self.lib.rt_film_upload(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0]),
                    "rt_film_upload")
        self.width, self.height = a.shape[1], a.shape[0]

    def film_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self.lib.rt_film_device_ptr(self.h, C.byref(p), C.byref(n)), "rt_film_device_ptr")
        return p.value, n.value

    def synchronize(self):
        self._check(self.lib.rt_synchronize(self.h), "rt_synchronize")

    # the film reduce of a multi-GPU render through the C ABI (RCCL inside the library)
    def comm_init_rank(self, unique_id: bytes, rank, n_ranks):
        """Joins the group named by `unique_id` (comm_unique_id() of rank 0) as `rank` of `n_ranks`:
        one process per GPU."""
        if len(unique_id) != _ffi.RT_COMM_ID_BYTES:
            raise ValueError(f"the id has {_ffi.RT_COMM_ID_BYTES} bytes")
        buf = C.create_string_buffer(bytes(unique_id), _ffi.RT_COMM_ID_BYTES)
        self._check(self.lib.rt_comm_init_rank(self.h, C.cast(buf, C.c_void_p), rank, n_ranks), "rt_comm_init_rank")

    def comm_destroy(self):
        self._check(self.lib.rt_comm_destroy(self.h), "rt_comm_destroy")

    def comm_info(self):
        r, n = C.c_int(), C.c_int()
        self._check(self.lib.rt_comm_info(self.h, C.byref(r), C.byref(n)), "rt_comm_info")
        return r.value, n.value

    def film_allreduce(self):
        """Sum of the films of the group this handle joined with comm_init_rank, in place, enqueued on
        the handle's stream."""
        arr = (C.c_void_p * 1)(self.h)
        self._check(self.lib.rt_film_allreduce(arr, 1), "rt_film_allreduce")

    def intersect_batch(self, o, d, time=None, tmin=0.001, tmax=float("inf"), origin_prim=None):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        fp = C.POINTER(C.c_float)
        tm = None
        if time is not None:
            time = np.ascontiguousarray(time, dtype=np.float32)
            tm = time.ctypes.data_as(fp)
        hit = np.empty(n, np.int32)
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        ip = C.POINTER(C.c_int32)
        frm = None
        if origin_prim is not None:
            origin_prim = np.ascontiguousarray(origin_prim, dtype=np.int32)
            frm = origin_prim.ctypes.data_as(ip)
        self._check(self.lib.rt_intersect_batch(self.h, n, o.ctypes.data_as(fp), d.ctypes.data_as(fp), tm, frm,
                                                C.c_float(tmin), C.c_float(tmax), hit.ctypes.data_as(ip),
                                                prim.ctypes.data_as(ip), t.ctypes.data_as(fp), u.ctypes.data_as(fp),
                                                v.ctypes.data_as(fp)), "rt_intersect_batch")
        return hit, prim, t, u, v

    def camera_rays(self, cam: RtCamera, pixel, stratum, seed=1):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        stratum = np.ascontiguousarray(stratum, dtype=np.uint32)
        n = pixel.shape[0]
        p = RtRenderParams()
        p.seed = seed
        o = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        t = np.empty(n, np.float32)
        fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        self._check(self.lib.rt_camera_rays(self.h, C.byref(cam), C.byref(p), n, pixel.ctypes.data_as(up),
                                            stratum.ctypes.data_as(up), o.ctypes.data_as(fp), d.ctypes.data_as(fp),
                                            t.ctypes.data_as(fp)), "rt_camera_rays")
        return o, d, t

    def eval_texture(self, tex, uvp):
        uvp = np.ascontiguousarray(uvp, dtype=np.float32).reshape(-1, 5)
        out = np.empty((uvp.shape[0], 3), np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.rt_eval_texture(self.h, tex, uvp.shape[0], uvp.ctypes.data_as(fp), out.ctypes.data_as(fp)),
                    "rt_eval_texture")
        return out

    def philox(self, ctr, key):
        ctr = np.ascontiguousarray(ctr, dtype=np.uint32).reshape(-1, 4)
        key = np.ascontiguousarray(key, dtype=np.uint32)
        out = np.empty_like(ctr)
        up = C.POINTER(C.c_uint32)
        self._check(self.lib.rt_philox(self.h, ctr.shape[0], ctr.ctypes.data_as(up), key.ctypes.data_as(up),
                                       out.ctypes.data_as(up)), "rt_philox")
        return out


def comm_unique_id() -> bytes:
    """The id rank 0 hands to its peers for Device.comm_init_rank (ncclGetUniqueId inside the library)."""
    buf = C.create_string_buffer(_ffi.RT_COMM_ID_BYTES)
    err = C.create_string_buffer(1024)
    if _ffi.lib.rt_comm_unique_id(C.cast(buf, C.c_void_p), err, 1024) < 0:
        raise RtError(err.value.decode())
    return buf.raw


def comm_init_all(devices):
    """One process driving several GPUs: forms one group of the given Devices (ncclCommInitAll)."""
    arr = (C.c_void_p * len(devices))(*[d.h for d in devices])
    if _ffi.lib.rt_comm_init_all(arr, len(devices)) < 0:
        raise RtError("rt_comm_init_all: " + _ffi.lib.rt_last_error(devices[0].h).decode())


def film_allreduce_all(devices):
    """The grouped all-reduce for the Devices of one comm_init_all group."""
    arr = (C.c_void_p * len(devices))(*[d.h for d in devices])
    if _ffi.lib.rt_film_allreduce(arr, len(devices)) < 0:
        raise RtError("rt_film_allreduce: " + _ffi.lib.rt_last_error(devices[0].h).decode())


def device_count():
    n = C.c_int()
    if _ffi.lib.rt_device_count(C.byref(n)) < 0:
        return 0
    return n.value
