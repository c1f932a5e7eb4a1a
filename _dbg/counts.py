import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["RT_AMD_LIBRARY"] = os.path.abspath("_dbg/librt_amd_dbg.so")
import torch
import raytracing_amd as ra
from raytracing_amd import _ffi
lib = _ffi.lib
names = {0: "node step", 1: "quad test", 2: "sphere test", 3: "medium test", 4: "round (busy lanes)", 5: "emit", 6: "camera", 7: "scatter",
         9: "shade iteration", 10: "metal", 11: "dielectric", 12: "lambert/iso", 13: "texture eval"}
wl = {"c3": (ra.scenes.cornell_book3, dict(width=600, spp=16), ra.RT_MODE_BOOK3),
      "c2": (ra.scenes.book1_final, dict(width=1200, spp=16), ra.RT_MODE_BOOK3),
      "c4": (ra.scenes.book2_final, dict(width=800, spp=16), ra.RT_MODE_BOOK3),
      "c5": (ra.scenes.stress, dict(n_prims=200000, width=960, height=540, spp=4, extent=58.0), ra.RT_MODE_BOOK3)}
for name in sys.argv[1:]:
    fn, kw, mode = wl[name]
    sb = ra.SceneBuilder(); cam = fn(sb, **kw)
    dev = ra.Device(0); dev.upload(sb); dev.film_resize(cam.image_width, cam.image_height)
    rc = cam.build()
    out = (ctypes.c_ulonglong * 64)()
    lib.rt_debug_counts(out, 1)
    st = dev.render(rc, mode=mode, seed=2)
    lib.rt_debug_counts(out, 1)
    print(name, "segments", st.segments, "camera rays", st.camera_rays, flush=True)
    for i, n in names.items():
        lanes, its = out[2 * i], out[2 * i + 1]
        if its:
            print(f"  {n:20s} wave-executions {its:12d}  per 64 segments {its / (st.segments / 64):7.2f}  lanes/execution {lanes / its:5.1f}")
