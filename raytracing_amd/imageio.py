"""Image files through the library's ImageWriter counterparts (export.rs:5-86): 8-bit RGB PNG
(linear values) and P3 PPM (gamma 2), both with the reference's truncating quantisation."""
import ctypes as C

import numpy as np

from . import _ffi


def _u8(img):
    a = np.ascontiguousarray(img)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("image must have shape (H, W, 3)")
    return a


def write_png(path, img):
    """img: (H, W, 3) float (linear, quantised here like PngWriter) or uint8 (written as is)."""
    a = _u8(img)
    if a.dtype == np.uint8:
        rc = _ffi.lib.rth_write_png_rgb8(str(path).encode(), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8)))
    else:
        a = np.ascontiguousarray(a, dtype=np.float32)
        rc = _ffi.lib.rth_write_png(str(path).encode(), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_float)))
    if rc < 0:
        raise OSError(f"cannot write {path}")


def write_ppm(path, img):
    """img: (H, W, 3) float (linear; gamma 2 applied here like PpmWriter) or uint8 (as is)."""
    a = _u8(img)
    if a.dtype == np.uint8:
        rc = _ffi.lib.rth_write_ppm_rgb8(str(path).encode(), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8)))
    else:
        a = np.ascontiguousarray(a, dtype=np.float32)
        rc = _ffi.lib.rth_write_ppm(str(path).encode(), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_float)))
    if rc < 0:
        raise OSError(f"cannot write {path}")


def read_png(path):
    """(H, W, 3) uint8 from an 8-bit RGB PNG -- the only kind ImageTexture::load accepts
    (texture.rs:95-122)."""
    w, h = C.c_uint32(), C.c_uint32()
    if _ffi.lib.rth_read_png_rgb8(str(path).encode(), C.byref(w), C.byref(h), None, 0) < 0:
        raise OSError(f"cannot read {path} as an 8-bit RGB PNG")
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    if _ffi.lib.rth_read_png_rgb8(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size) < 0:
        raise OSError(f"cannot read {path}")
    return out
