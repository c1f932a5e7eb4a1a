#!/usr/bin/env python3
"""Timing probe for the render path (GPU only): renders one workload with several launch
settings and prints Mrays/s for each.  usage: perf_probe.py [workload] [spp]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401
import raytracing_amd as ra  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
fn, kw, mode, _ = WORKLOADS[name]
kw = dict(kw)
if len(sys.argv) > 2:
    kw["spp"] = int(sys.argv[2])
sb = ra.SceneBuilder()
cam = fn(sb, **kw)
dev = ra.Device(0)
dev.upload(sb)
rc = cam.build()
dev.film_resize(cam.image_width, cam.image_height)
dev.render(rc, mode=mode, seed=1)
for accel in (("reference", 4), ("sah", 4), ("reference", 2)):
    dev.set_accel(*accel)
    t_up = time.perf_counter()
    dev.upload(sb)
    t_up = time.perf_counter() - t_up
    info = dev.accel_info()
    for v in (dict(wavefront=True), dict()):
        dev.render(rc, mode=mode, seed=2, **v)
        t0 = time.perf_counter()
        st = dev.render(rc, mode=mode, seed=2, **v)
        dt = time.perf_counter() - t0
        print(f"{name} {info['kind']} {accel[0]:9s} {'wavefront ' if v else 'persistent'}: {st.segments / dt / 1e6:9.1f} Mrays/s"
              f"  {dt * 1e3:8.2f} ms  segs={st.segments} nodes={info['nodes']} stack={info['stack_entries']} cost={info['cost']:.1f} upload={t_up * 1e3:.0f}ms", flush=True)
    if info["kind"] == "flat":
        break
