#!/usr/bin/env python3
"""Segment rate of a workload under each variant of the persistent kernel (lane-bound paths, a pool of paths
per wave) and the ray-queue pipeline.  usage: pipeline_probe.py [name[:spp] ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401
import raytracing_amd as ra  # noqa: E402
from bench import WORKLOADS  # noqa: E402

DEFAULT_SPP = {"c1": 20, "c2": 36, "c3": 64, "c4": 64, "c5": 4}
dev = ra.Device(0)
for arg in sys.argv[1:] or ["c2", "c4"]:
    name, _, spp = arg.partition(":")
    fn, kw, mode, _ = WORKLOADS[name]
    kw = dict(kw, spp=int(spp) if spp else DEFAULT_SPP[name])
    sb = ra.SceneBuilder()
    cam = fn(sb, **kw)
    dev.upload(sb)
    dev.film_resize(cam.image_width, cam.image_height)
    rc = cam.build()
    for label, args in (("lane-bound", dict(regroup=False)), ("pool per wave", dict(regroup=True)), ("ray queues", dict(wavefront=True))):
        dev.render(rc, mode=mode, seed=1, **args)
        best = None
        for rep in range(3):
            st = dev.render(rc, mode=mode, seed=2, time_kernels=True, **args)
            ms = st.render_ms
            best = ms if best is None else min(best, ms)
        print(f"{name} {label:14s}: {st.segments / best / 1e6:8.3f} G segments/s  {best:9.3f} ms  regrouped={st.regrouped}", flush=True)
