"""MI355X-native path-tracing backend for the render hot path of piemot/raytracing.

The compute path is librt_amd.so: hand-written HIP kernels for gfx950 behind the C ABI of
include/rt_amd.h, plus the C++ host layer that mirrors the reference's scene/camera/config API.
This package only binds it; importing fails if the library has not been built.
"""
from ._ffi import RT_FLAG_TIME_KERNELS, RT_FLAG_WAVEFRONT, RT_MODE_BOOK3, RT_MODE_REF_EXACT  # noqa: F401
from .api import (CameraSettings, Device, RtError, SceneBuilder, Stats, device_count,  # noqa: F401
                  comm_init_all, comm_unique_id, film_allreduce_all,
                  SHAPE_DISC, SHAPE_PARALLELOGRAM, SHAPE_TRIANGLE)
from . import imageio, progressive, scenes  # noqa: F401
