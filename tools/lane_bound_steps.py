#!/usr/bin/env python3
"""Step statistics of the lane-bound persistent kernel behind the wide tree: how many node steps,
primitive steps and shading passes a wave runs and with how many lanes each.  Needs the
instrumented library (make stats-lib):

    RT_AMD_LIBRARY=$PWD/raytracing_amd/librt_amd_stats.so python tools/lane_bound_steps.py [c2|c4|c5]
"""
import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401
import raytracing_amd as ra
from bench import WORKLOADS
dev = ra.Device(0)
for name in (sys.argv[1:] or ["c2", "c4"]):
    fn, kw, mode, _ = WORKLOADS[name]
    kw = dict(kw, spp={"c2": 36, "c4": 64, "c5": 4}[name])
    sb = ra.SceneBuilder(); cam = fn(sb, **kw)
    rc = cam.build()
    for builder, post in (("auto", True), ("auto", False), ("reference", True), ("reference", False)):
        dev.set_accel(builder, 4, post_tests=post)
        dev.upload(sb)
        info = dev.accel_info()
        dev.film_resize(cam.image_width, cam.image_height)
        NL = 8
        out = (C.c_ulonglong * (3 * NL))()
        dev.lib.rt_debug_regroup_stats(out)
        st = dev.render(rc, mode=mode, seed=2, regroup=False, time_kernels=True)
        assert dev.lib.rt_debug_regroup_stats(out) == 0
        names = ["node steps", "primitive steps", "shading passes (lanes with a hit)", "  lanes that scatter", "  lanes taking a sample", "calls for rounds (lanes traversing)"]
        seg = st.segments
        print(f"== {name} {builder} post tests {post} ({info['post_tests']}), {info['nodes']} nodes, {info['box_leaves']} box leaves: {seg} segments, {st.kernel_ms[4]:.3f} ms")
        for i, n in enumerate(names):
            t, l = out[i], out[NL + i]
            print(f"{n:38s} steps {t:12d}  lanes/step {l / max(t, 1):6.2f}  lane-steps per segment {l / seg:7.3f}  wave-steps per 64 segments {64 * t / seg:7.3f}")
        # clock ticks (100 MHz) the waves spent per region of the loop, as shares
        ticks = {"rounds of the walk": out[2 * NL + 5], "post tests + the hit's own contribution": out[2 * NL + 2], "new samples + Philox block": out[2 * NL + 6],
                 "camera rays": out[2 * NL + 4], "scatter": out[2 * NL + 3]}
        total = sum(ticks.values()) or 1
        print("   time shares: " + ", ".join(f"{k} {v / total:.1%}" for k, v in ticks.items()))
        if info["post_tests"] == 0 and post:
            break
