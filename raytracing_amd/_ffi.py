"""ctypes binding of librt_amd.so (include/rt_amd.h).

The library is the product: HIP kernels for gfx950 plus the C++ host scene layer.  There is
no Python or CPU fallback; if the shared object is missing, importing this module raises.
"""
import ctypes as C
import os
from pathlib import Path

# PyTorch ships its own HIP/HSA runtime (soname libamdhip64.so.7).  Two HIP runtimes in one
# process leave the second without a device, so torch is loaded first and librt_amd.so then
# binds to the runtime already in the process.
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)

# RT_AMD_LIBRARY names another build of the same library (kernel experiments)
_LIB_PATH = Path(os.environ.get("RT_AMD_LIBRARY") or Path(__file__).resolve().parent / "librt_amd.so")

u32, u64, i32, f32, f64 = C.c_uint32, C.c_uint64, C.c_int32, C.c_float, C.c_double
P = C.POINTER


class RtNode(C.Structure):
    _fields_ = [("lmin", f32 * 3), ("left", u32), ("lmax", f32 * 3), ("right", u32),
                ("rmin", f32 * 3), ("count", u32), ("rmax", f32 * 3), ("pad", u32)]


class RtQuad(C.Structure):
    _fields_ = [("q", f32 * 3), ("d", f32), ("n", f32 * 3), ("area", f32), ("avec", f32 * 3), ("shape", u32),
                ("bvec", f32 * 3), ("mat", u32), ("u", f32 * 3), ("prim_id", u32), ("v", f32 * 3), ("pad", u32)]


class RtSphere(C.Structure):
    _fields_ = [("c0", f32 * 3), ("radius", f32), ("cv", f32 * 3), ("mat", u32),
                ("prim_id", u32), ("moving", u32), ("u_shift", f32), ("pad", u32)]


class RtMedium(C.Structure):
    _fields_ = [("first", u32), ("count", u32), ("neg_inv_density", f32), ("mat", u32),
                ("prim_id", u32), ("pad0", u32), ("pad1", u32), ("pad2", u32)]


class RtMaterial(C.Structure):
    _fields_ = [("type", u32), ("tex", u32), ("albedo", f32 * 3), ("param", f32), ("needs_uv", u32), ("pad", u32)]


class RtTexture(C.Structure):
    _fields_ = [("type", u32), ("color", f32 * 3), ("scale", f32), ("a", u32), ("b", u32), ("offset", u32)]


class RtSceneDesc(C.Structure):
    _fields_ = [("n_nodes", u32), ("n_refs", u32), ("n_quads", u32), ("n_spheres", u32), ("n_media", u32),
                ("n_materials", u32), ("n_textures", u32), ("n_lights", u32), ("n_texel_bytes", u32),
                ("n_noise", u32), ("root", u32), ("lights_kind", u32), ("n_prims", u32), ("max_stack", u32),
                ("nodes", P(RtNode)), ("refs", P(u32)), ("quads", P(RtQuad)), ("spheres", P(RtSphere)),
                ("media", P(RtMedium)), ("materials", P(RtMaterial)), ("textures", P(RtTexture)),
                ("light_refs", P(u32)), ("texels", P(C.c_uint8)), ("noise_vec", P(f32)), ("noise_perm", P(i32))]


class RtCamera(C.Structure):
    _fields_ = [("image_width", u32), ("image_height", u32), ("sqrt_spp", u32), ("max_depth", u32),
                ("background_sky", u32), ("antialiasing", u32), ("background", f64 * 3), ("center", f64 * 3),
                ("pixel_00", f64 * 3), ("pxdelta_u", f64 * 3), ("pxdelta_v", f64 * 3), ("defocus_angle", f64),
                ("defocus_disk_u", f64 * 3), ("defocus_disk_v", f64 * 3)]


RT_ABI_VERSION = 3  # of include/rt_amd.h as this file mirrors it; checked against the library at load


class _Sized(C.Structure):
    """A struct of rt_amd.h that starts with struct_size: set to this binding's size on creation."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(self)


class RtRenderParams(_Sized):
    _fields_ = [("struct_size", u32), ("mode", u32), ("flags", u32), ("reserved", u32), ("seed", u64),
                ("stratum_begin", u32), ("stratum_end", u32),
                ("pool_size", u32), ("time_every", u32), ("refill_min", u32), ("waves_per_simd", u32)]


class RtStats(_Sized):
    _fields_ = [("struct_size", u32), ("reserved", u32),
                ("segments", u64), ("camera_rays", u64), ("dropped", u64), ("iterations", u64),
                ("launches", u64 * 5), ("timed_launches", u64 * 5), ("kernel_ms", f64 * 5), ("render_ms", f64),
                ("regrouped", u64)]


class CameraBuilderStruct(C.Structure):
    """Field layout shared by rth_camera_builder and the oracle's camera struct."""
    _fields_ = [("image_width", u32), ("image_height", u32), ("vfov_deg", f64), ("samples_per_px", u32),
                ("max_depth", u32), ("background_sky", i32), ("antialiasing", i32), ("background", f64 * 3), ("center", f64 * 3),
                ("target", f64 * 3), ("vup", f64 * 3), ("defocus_angle_deg", f64), ("focal_length", f64)]


RT_MODE_REF_EXACT, RT_MODE_BOOK3 = 0, 1


class RtAccelInfo(_Sized):
    _fields_ = [("struct_size", u32), ("kind", u32), ("nodes", u32), ("depth", u32), ("stack_entries", u32), ("cost", f64),
                ("flat_quads", u32), ("flat_boxes", u32), ("flat_alone_spheres", u32), ("node_bytes", u32),
                ("flat_boxes_aligned", u32), ("flat_boxes_upright", u32), ("flat_rects", u32), ("box_leaves", u32), ("leaves", u32),
                ("post_tests", u32), ("reserved", u32), ("walk_cost", f64)]


RT_BUILD_REFERENCE, RT_BUILD_SAH, RT_BUILD_AUTO = 0, 1, 2
RT_BUILD_FP32_BOXES, RT_BUILD_BYTE_BOXES, RT_BUILD_NO_BOX_LEAVES, RT_BUILD_NO_POST_TESTS = 0x100, 0x200, 0x400, 0x800
RT_FLAG_TIME_KERNELS = 1
RT_FLAG_WAVEFRONT = 2
RT_FLAG_REGROUP = 4
RT_FLAG_NO_REGROUP = 8
RT_FLAG_GENERAL_KERNELS = 16
RT_VARIANT_REGROUPED, RT_VARIANT_TIMELESS, RT_VARIANT_BOXED, RT_VARIANT_PLAIN, RT_VARIANT_EXTRAS, RT_VARIANT_SIMPLE = 1, 2, 4, 8, 16, 32
RT_KIND_NODE, RT_KIND_QUAD, RT_KIND_SPHERE, RT_KIND_MEDIUM = 0, 1, 2, 3
RT_COMM_ID_BYTES = 128
RT_NODE_LEAF = 0xFFFFFFFF

# name -> (restype, argtypes); every symbol include/rt_amd.h declares
_vp = C.c_void_p
_d3 = P(f64)
SIGNATURES = {
    "rt_abi_version": (u32, []),
    "rt_last_kernel_variant": (u32, [_vp]),
    "rt_device_count": (C.c_int, [P(C.c_int)]),
    "rt_create": (C.c_int, [C.c_int, P(_vp)]),
    "rt_destroy": (None, [_vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_set_stream": (C.c_int, [_vp, _vp]),
    "rt_synchronize": (C.c_int, [_vp]),
    "rt_upload_scene": (C.c_int, [_vp, P(RtSceneDesc)]),
    "rt_render": (C.c_int, [_vp, P(RtCamera), P(RtRenderParams)]),
    "rt_film_clear": (C.c_int, [_vp]),
    "rt_film_resize": (C.c_int, [_vp, u32, u32]),
    "rt_film_bind": (C.c_int, [_vp, _vp, u32, u32]),
    "rt_film_device_ptr": (C.c_int, [_vp, P(_vp), P(u64)]),
    "rt_set_accel": (C.c_int, [_vp, u32, u32]),
    "rt_get_accel_info": (C.c_int, [_vp, P(RtAccelInfo)]),
    "rt_accel_selftest": (C.c_int, [P(RtSceneDesc), u32, P(RtAccelInfo), C.c_char_p, u32]),
    "rt_film_download": (C.c_int, [_vp, P(f32), f32]),
    "rt_film_upload": (C.c_int, [_vp, P(f32), u32, u32]),
    "rt_film_quantize": (C.c_int, [_vp, P(C.c_uint8), f64, u32]),
    "rt_film_quantize_noalloc": (C.c_int, [_vp, P(C.c_uint8), f64, u32]),
    "rt_get_stats": (C.c_int, [_vp, P(RtStats)]),
    "rt_comm_init_all": (C.c_int, [P(_vp), C.c_int]),
    "rt_comm_unique_id": (C.c_int, [_vp, C.c_char_p, u32]),
    "rt_comm_init_rank": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "rt_comm_destroy": (C.c_int, [_vp]),
    "rt_comm_info": (C.c_int, [_vp, P(C.c_int), P(C.c_int)]),
    "rt_film_allreduce": (C.c_int, [P(_vp), C.c_int]),
    "rt_intersect_batch": (C.c_int, [_vp, u32, P(f32), P(f32), P(f32), P(i32), f32, f32, P(i32), P(i32), P(f32), P(f32), P(f32)]),
    "rt_camera_rays": (C.c_int, [_vp, P(RtCamera), P(RtRenderParams), u32, P(u32), P(u32), P(f32), P(f32), P(f32)]),
    "rt_eval_texture": (C.c_int, [_vp, u32, u32, P(f32), P(f32)]),
    "rt_philox": (C.c_int, [_vp, u32, P(u32), P(u32), P(u32)]),
    "rth_camera_defaults": (None, [P(CameraBuilderStruct)]),
    "rth_camera_with_aspect_ratio": (C.c_int, [P(CameraBuilderStruct), u32, f64]),
    "rth_camera_build": (C.c_int, [P(CameraBuilderStruct), P(RtCamera), C.c_char_p, u32]),
    "rth_scene_new": (_vp, []),
    "rth_scene_free": (None, [_vp]),
    "rth_last_error": (C.c_char_p, [_vp]),
    "rth_tex_solid": (C.c_int, [_vp, f64, f64, f64]),
    "rth_tex_checker": (C.c_int, [_vp, f64, C.c_int, C.c_int]),
    "rth_tex_image": (C.c_int, [_vp, u32, u32, P(C.c_uint8)]),
    "rth_tex_image_png": (C.c_int, [_vp, C.c_char_p]),
    "rth_tex_noise": (C.c_int, [_vp, f64, _d3, P(i32)]),
    "rth_mat_lambertian": (C.c_int, [_vp, C.c_int]),
    "rth_mat_light": (C.c_int, [_vp, C.c_int]),
    "rth_mat_metal": (C.c_int, [_vp, f64, f64, f64, f64]),
    "rth_mat_dielectric": (C.c_int, [_vp, f64]),
    "rth_mat_isotropic": (C.c_int, [_vp, C.c_int]),
    "rth_sphere": (C.c_int, [_vp, _d3, _d3, f64, C.c_int]),
    "rth_quad": (C.c_int, [_vp, C.c_int, _d3, _d3, _d3, C.c_int]),
    "rth_box3": (C.c_int, [_vp, _d3, _d3, C.c_int]),
    "rth_list": (C.c_int, [_vp, P(C.c_int), C.c_int]),
    "rth_bvh": (C.c_int, [_vp, P(C.c_int), C.c_int]),
    "rth_translate": (C.c_int, [_vp, C.c_int, _d3]),
    "rth_rotate_y": (C.c_int, [_vp, C.c_int, f64]),
    "rth_medium": (C.c_int, [_vp, C.c_int, f64, C.c_int]),
    "rth_set_world": (C.c_int, [_vp, C.c_int]),
    "rth_set_lights": (C.c_int, [_vp, C.c_int]),
    "rth_prim_count": (C.c_int, [_vp]),
    "rth_load_toml": (C.c_int, [_vp, C.c_char_p]),
    "rth_load_scene_toml": (C.c_int, [_vp, C.c_char_p, C.c_char_p, P(CameraBuilderStruct), P(C.c_int)]),
    "rth_list_add": (C.c_int, [_vp, C.c_int, C.c_int]),
    "rth_material_by_name": (C.c_int, [_vp, C.c_char_p]),
    "rth_flatten": (C.c_int, [_vp, P(P(RtSceneDesc))]),
    "rth_bvh_stats": (C.c_int, [_vp, C.c_int, P(C.c_int), P(C.c_int)]),
    "rth_color_hex": (None, [u32, _d3]),
    "rth_color_quantize": (None, [_d3, C.c_int, P(C.c_uint8)]),
    "rth_sphere_uv": (None, [_d3, _d3]),
    "rth_interval_contains": (C.c_int, [f64, f64, f64]),
    "rth_interval_surrounds": (C.c_int, [f64, f64, f64]),
    "rth_interval_overlap": (C.c_int, [f64, f64, f64, f64, _d3]),
    "rth_interval_expand": (None, [f64, f64, f64, _d3]),
    "rth_interval_clamp": (f64, [f64, f64, f64]),
    "rth_write_png": (C.c_int, [C.c_char_p, u32, u32, P(f32)]),
    "rth_write_ppm": (C.c_int, [C.c_char_p, u32, u32, P(f32)]),
    "rth_write_png_rgb8": (C.c_int, [C.c_char_p, u32, u32, P(C.c_uint8)]),
    "rth_write_ppm_rgb8": (C.c_int, [C.c_char_p, u32, u32, P(C.c_uint8)]),
    "rth_read_png_rgb8": (C.c_int, [C.c_char_p, P(u32), P(u32), P(C.c_uint8), u64]),
}


def load_library(path=None):
    path = Path(path) if path else _LIB_PATH
    if not path.exists():
        raise ImportError(
            f"{path} is missing: build it with `make` at the repository root "
            "(hipcc --offload-arch=gfx950).  raytracing_amd has no CPU or pure-Python path.")
    lib = C.CDLL(os.fspath(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the build is stale
        fn.restype = res
        fn.argtypes = args
    have = lib.rt_abi_version()
    if have != RT_ABI_VERSION:
        raise ImportError(f"{path} speaks ABI {have}, this binding ABI {RT_ABI_VERSION}: rebuild the library (`make`)")
    return lib


lib = load_library()
