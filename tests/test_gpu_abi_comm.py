"""C ABI on the device: the struct-size handshake of rt_render and the film reduce through RCCL
(rt_comm_* / rt_film_allreduce).  A one-GPU box can form groups of one rank only -- RCCL takes
one rank per GPU -- so what is checked here is the whole call sequence of both ways to form a
group, the ordering of the reduce behind the render on the handle's stream, and every refusal;
a group of one returns the film unchanged, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import _ffi

lib = _ffi.lib


def _scene(dev, width=64, spp=9):
    sb = ra.SceneBuilder()
    cam = ra.scenes.cornell_book3(sb, width=width, spp=spp)
    dev.upload(sb)
    dev.film_resize(width, width)
    return cam


@pytest.mark.gpu
def test_render_params_from_another_header_are_refused():
    dev = ra.Device(0)
    try:
        cam = _scene(dev)
        rc = cam.build()
        p = _ffi.RtRenderParams()
        p.mode, p.seed = ra.RT_MODE_BOOK3, 3
        assert lib.rt_render(dev.h, C.byref(rc), C.byref(p)) == 0
        want = dev.film()
        # struct_size 0: a binding from before the field existed
        p.struct_size = 0
        assert lib.rt_render(dev.h, C.byref(rc), C.byref(p)) < 0
        assert "struct_size" in lib.rt_last_error(dev.h).decode()
        # a newer header than the library: refused, its extra fields cannot be honoured
        p.struct_size = C.sizeof(p) + 8
        assert lib.rt_render(dev.h, C.byref(rc), C.byref(p)) < 0
        assert "newer header" in lib.rt_last_error(dev.h).decode()
        # an older header that ends after stratum_end: the fields it lacks take their defaults
        p.struct_size = _ffi.RtRenderParams.pool_size.offset
        p.waves_per_simd = 77  # beyond what the library may read
        dev.film_clear()
        assert lib.rt_render(dev.h, C.byref(rc), C.byref(p)) == 0, lib.rt_last_error(dev.h).decode()
        assert np.allclose(dev.film(), want, rtol=2e-5, atol=1e-6)
        # both regrouping flags: no precedence to guess
        p = _ffi.RtRenderParams()
        p.mode, p.flags = ra.RT_MODE_BOOK3, _ffi.RT_FLAG_REGROUP | _ffi.RT_FLAG_NO_REGROUP
        assert lib.rt_render(dev.h, C.byref(rc), C.byref(p)) < 0
        assert "both set" in lib.rt_last_error(dev.h).decode()
        # stats into a shorter struct: filled up to the caller's size
        st = _ffi.RtStats()
        st.struct_size = _ffi.RtStats.iterations.offset
        st.iterations = 4242
        assert lib.rt_get_stats(dev.h, C.byref(st)) == 0
        assert st.iterations == 4242 and st.struct_size == _ffi.RtStats.iterations.offset
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_film_allreduce_through_the_c_abi_with_a_group_of_one():
    dev = ra.Device(0)
    try:
        cam = _scene(dev)
        rc = cam.build()
        # no group yet: refused, nothing hangs
        arr = (C.c_void_p * 1)(dev.h)
        assert lib.rt_film_allreduce(arr, 1) < 0 and "in no group" in lib.rt_last_error(dev.h).decode()
        assert dev.comm_info() == (0, 0)

        # one process, one handle per GPU: ncclCommInitAll
        ra.comm_init_all([dev])
        assert dev.comm_info() == (0, 1)
        with pytest.raises(ra.RtError, match="already belongs"):
            ra.comm_init_all([dev])
        for seed in (3, 4, 5):  # frames in a row: clear, render, reduce on one stream
            dev.film_clear()
            st = dev.render(rc, mode=ra.RT_MODE_BOOK3, seed=seed)
            before = dev.film()
            ra.film_allreduce_all([dev])
            dev.synchronize()
            after = dev.film()
            assert st.segments > 0 and before.sum() > 0
            assert np.array_equal(before.view(np.uint32), after.view(np.uint32))  # bit for bit
        # a second handle on the same GPU cannot join a group with the first
        dev2 = ra.Device(0)
        try:
            dev.comm_destroy()
            assert dev.comm_info() == (0, 0)
            with pytest.raises(ra.RtError, match="two handles on GPU 0"):
                ra.comm_init_all([dev, dev2])
        finally:
            dev2.close()

        # one process per GPU: rank 0 makes the id, every rank joins with it
        uid = ra.comm_unique_id()
        assert len(uid) == _ffi.RT_COMM_ID_BYTES and any(uid)
        with pytest.raises(ra.RtError, match="rank outside"):
            dev.comm_init_rank(uid, 1, 1)
        dev.comm_init_rank(uid, 0, 1)
        assert dev.comm_info() == (0, 1)
        dev.film_clear()
        dev.render(rc, mode=ra.RT_MODE_BOOK3, seed=6)
        before = dev.film()
        dev.film_allreduce()
        after = dev.film()  # the download runs on the handle's stream, behind the reduce
        assert np.array_equal(before.view(np.uint32), after.view(np.uint32))
        # the films of a group must agree in size: checked before anything is enqueued (here: no film)
        dev3 = ra.Device(0)
        try:
            arr3 = (C.c_void_p * 1)(dev3.h)
            assert lib.rt_film_allreduce(arr3, 1) < 0
        finally:
            dev3.close()
        dev.comm_destroy()
        dev.comm_destroy()  # leaving twice is fine
    finally:
        dev.close()
