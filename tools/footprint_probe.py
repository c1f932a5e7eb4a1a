#!/usr/bin/env python3
"""Segment rate of the stress scene against its size at constant primitive density (extent grows
with the cube root of the count): separates the cost of a deeper tree from the cost of a working
set that no longer fits the L2.  usage: python tools/footprint_probe.py [counts...]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raytracing_amd as ra  # noqa: E402


W, H, SPP = 3840, 2160, 4


def main():
    counts = [int(a) for a in sys.argv[1:]] or [31250, 125000, 500000, 1000000]
    dev = ra.Device(0)
    for n in counts:
        sb = ra.SceneBuilder()
        t0 = time.perf_counter()
        cam = ra.scenes.stress(sb, n_prims=n, width=W, height=H, spp=SPP, extent=100.0 * (n / 1e6) ** (1.0 / 3.0))
        for fp32_boxes in (True, False):  # 128-byte nodes, 64-byte quantised nodes
            dev.set_accel("auto", 4, fp32_boxes=fp32_boxes)
            dev.upload(sb)
            t1 = time.perf_counter()
            dev.film_resize(cam.image_width, cam.image_height)
            rc = cam.build()
            dev.render(rc, mode=ra.RT_MODE_BOOK3, seed=1)
            ms = min(dev.render(rc, mode=ra.RT_MODE_BOOK3, seed=2, time_kernels=True).kernel_ms[4] for _ in range(3))
            st = dev.stats()
            info = dev.accel_info()
            print(f"n={n:8d} node_bytes={info['node_bytes']:3d} nodes={info['nodes']:7d} depth={info['depth']:2d} stack={info['stack_entries']:2d} cost={info['cost']:7.1f} "
                  f"segments={st.segments:9d} seg/ray={st.segments / st.camera_rays:5.2f} {ms:8.3f} ms {st.segments / ms / 1e6:7.3f} G/s "
                  f"(build+upload {t1 - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
