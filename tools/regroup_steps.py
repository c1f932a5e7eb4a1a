#!/usr/bin/env python3
"""Per-step statistics of the regrouping persistent kernel: turns, slots per turn and share of the
waves' time for each kind of step.  Behind a tree the ray step is a traversal round of the lanes holding
a ray, and harvest / refill move hits and rays between lanes and the pool.  Needs the instrumented library (make stats-lib):

    RT_AMD_LIBRARY=$PWD/raytracing_amd/librt_amd_stats.so python tools/regroup_steps.py [c1|c3|c2|c4|c5]
"""
import sys, ctypes as C, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import raytracing_amd as ra
from bench import WORKLOADS
dev = ra.Device(0)
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c3"
fn, kw, mode, _ = WORKLOADS[name]
kw = dict(kw, spp={"c1": 20, "c3": 64, "c2": 36, "c4": 64, "c5": 4}[name])
sb = ra.SceneBuilder(); cam = fn(sb, **kw); dev.upload(sb)
dev.film_resize(cam.image_width, cam.image_height); rc = cam.build()
NL = 8
out = (C.c_ulonglong * (3 * NL))()
dev.lib.rt_debug_regroup_stats(out)
st = dev.render(rc, mode=mode, seed=2, regroup=True, time_kernels=True)
assert dev.lib.rt_debug_regroup_stats(out) == 0
names = ["free(camera)", "ray(trace)", "diffuse", "metal", "glass", "end", "harvest", "refill"]  # the last two: k_paths_pool only
tot_turns = sum(out[:NL]); tot_ticks = sum(out[2*NL:3*NL])
for i, n in enumerate(names):
    t, l, k = out[i], out[NL + i], out[2 * NL + i]
    print(f"{n:14s} turns {t:12d} ({100*t/tot_turns:5.1f}%) lanes/turn {l/max(t,1):6.2f}  ticks/turn {k/max(t,1):8.1f}  time share {100*k/tot_ticks:5.1f}%")
print("ms", st.kernel_ms[4], "segments", st.segments)
