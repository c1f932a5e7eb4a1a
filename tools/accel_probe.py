#!/usr/bin/env python3
"""Segment rate of one workload under every search structure the upload can build: builder
(reference grouping / surface-area regrouping), box leaves on or off.  usage: accel_probe.py [name[:spp] ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401
import raytracing_amd as ra  # noqa: E402
from bench import WORKLOADS  # noqa: E402

DEFAULT_SPP = {"c1": 20, "c2": 36, "c3": 64, "c4": 64, "c5": 4}
dev = ra.Device(0)
for arg in sys.argv[1:] or ["c4"]:
    name, _, spp = arg.partition(":")
    fn, kw, mode, _ = WORKLOADS[name]
    kw = dict(kw, spp=int(spp) if spp else DEFAULT_SPP[name])
    sb = ra.SceneBuilder()
    cam = fn(sb, **kw)
    rc = cam.build()
    for builder in ("reference", "sah", "auto"):
        for leaves in (True, False):
            dev.set_accel(builder, 4, box_leaves=leaves)
            dev.upload(sb)
            info = dev.accel_info()
            dev.film_resize(cam.image_width, cam.image_height)
            dev.render(rc, mode=mode, seed=1)
            best = None
            for rep in range(3):
                st = dev.render(rc, mode=mode, seed=2, time_kernels=True)
                ms = st.kernel_ms[4]
                best = ms if best is None else min(best, ms)
            print(f"{name} {builder:9s} box leaves {'on ' if leaves else 'off'}: {st.segments / best / 1e6:8.3f} G segments/s  {best:9.3f} ms  "
                  f"nodes={info['nodes']} leaves={info['leaves']} box_leaves={info['box_leaves']} stack={info['stack_entries']} cost={info['cost']:.2f}", flush=True)
