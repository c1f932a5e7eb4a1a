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
variants = [dict(wavefront=True)]
for wps in (4,):
    for rm in (8,):
        variants.append(dict(refill_min=rm, waves_per_simd=wps))
dev.render(rc, mode=mode, seed=1)
for v in variants:
    t0 = time.perf_counter()
    st = dev.render(rc, mode=mode, seed=2, **v)
    dt = time.perf_counter() - t0
    print(f"{name} {v}: {st.segments / dt / 1e6:9.1f} Mrays/s  {dt * 1e3:8.2f} ms  segs={st.segments}", flush=True)
