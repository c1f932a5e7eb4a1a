import sys, time
sys.path.insert(0, '.')
import numpy as np
import raytracing_amd as ra

dev = ra.Device(0)
for name, fn, mode, kw in [("C1", ra.scenes.cornell_main, 0, dict(width=600, spp=16)),
                           ("C1big", ra.scenes.cornell_main, 0, dict(width=1200, spp=64)),
                           ("C3", ra.scenes.cornell_book3, 1, dict(width=1200, spp=64))]:
    sb = ra.SceneBuilder()
    cam = fn(sb, **kw)
    dev.upload(sb)
    rc = cam.build()
    for rep in range(3):
        dev.film_resize(cam.image_width, cam.image_height)
        t0 = time.time()
        st = dev.render(rc, mode=mode, seed=1, time_kernels=(rep == 2))
        dt = time.time() - t0
        print(name, rep, "wall %.4f s render_ms %.2f segs %d  Mseg/s %.1f iters %d" % (dt, st.render_ms, st.segments, st.segments / st.render_ms / 1e3, st.iterations),
              "kernel_ms", [round(x, 2) for x in st.kernel_ms], "dropped", st.dropped)
    img = dev.film(1.0 / cam.effective_spp)
    print(name, "mean", img.mean(axis=(0, 1)))
