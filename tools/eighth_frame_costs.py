#!/usr/bin/env python3
"""What one of eight GPUs does per frame of the headline workload, priced on ONE GPU.

Eight ranks split the 961 strata of the Book-3 Cornell frame 121 / 120 each; a rank's frame is: clear
the film, launch the path kernel for its strata, all-reduce the 17 MB film, host sync.  This tool runs
exactly that through DistributedRenderer with a process group of ONE rank over RCCL (the all-reduce
then goes through RCCL's launch path but moves nothing over xGMI) and times the parts with events on
the renderer's stream, frame after frame, once waiting for every frame (render) and once with two
films in turn (render_async).  It ends with a PREDICTION of the eight-GPU scaling efficiency from
these parts plus an assumed eight-rank all-reduce time (argument --allreduce-ms, default 0.3: a ring
over seven xGMI links moves 2 * 7/8 * 17.3 MB = 30 MB per rank at ~100 GB/s) -- a prediction, not a
measurement: the node with eight GPUs is the driver's.

    python tools/eighth_frame_costs.py [--frames 20] [--allreduce-ms 0.3] [--reduce torch|c-abi]
"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import raytracing_amd as ra  # noqa: E402
from raytracing_amd import distributed as rdist  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--allreduce-ms", type=float, default=0.3)
ap.add_argument("--reduce", default="torch", choices=["torch", "c-abi"])
args = ap.parse_args()

os.environ.setdefault("MASTER_PORT", str(rdist.free_port()))
rank, world, local = rdist.init_from_env("nccl", single_rank_group=True)
assert world == 1
sb = ra.SceneBuilder()
cam = ra.scenes.cornell_book3(sb, width=1200, spp=1000)
dev = ra.Device(0)
dev.upload(sb)
rc = cam.build()
n_strata = cam.effective_spp
renderer = rdist.DistributedRenderer(dev, 0, reduce=args.reduce)
renderer.bind_film(cam.image_width, cam.image_height)
mode = ra.RT_MODE_BOOK3


def ev():
    return torch.cuda.Event(enable_timing=True)


def frame_parts(strata, frames):
    """Per frame, every part waited for: clear, render (wall and kernel), all-reduce, in ms."""
    out = []
    film = renderer.films[0]
    for f in range(frames + 2):
        with torch.cuda.stream(renderer.stream):
            e0, e1, e2, e3 = ev(), ev(), ev(), ev()
            e0.record()
            film.zero_()
            e1.record()
            dev.film_bind(film.data_ptr(), cam.image_width, cam.image_height)
            t0 = time.perf_counter()
            st = dev.render(rc, mode=mode, seed=1 + f, stratum_begin=0, stratum_end=strata, time_kernels=True)
            wall = (time.perf_counter() - t0) * 1e3
            e2.record()
            if args.reduce == "c-abi":
                renderer._join_c_abi_group()
                dev.film_allreduce()
            else:
                dist.all_reduce(film)
            e3.record()
        t1 = time.perf_counter()
        renderer.stream.synchronize()
        sync = (time.perf_counter() - t1) * 1e3
        if f >= 2:
            out.append((e0.elapsed_time(e1), wall, st.kernel_ms[4], e2.elapsed_time(e3), sync, st.segments))
    return np.array(out)


def frames_per_second(strata, frames, overlapped):
    renderer.wait()
    fn = renderer.render_async if overlapped else renderer.render
    for f in range(2):
        fn(rc, strata, 0, 1, mode, seed=100 + f)
    renderer.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        fn(rc, strata, 0, 1, mode, seed=200 + f)
    renderer.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / frames


eighth = (n_strata + 7) // 8  # 121: the largest share
print(f"Book-3 Cornell box {cam.image_width}x{cam.image_height}, {n_strata} strata; one GPU, process group of one rank over RCCL, film reduce: {args.reduce}")
for name, strata in (("whole frame (1 GPU)", n_strata), (f"an eighth ({eighth} strata)", eighth)):
    p = frame_parts(strata, args.frames if strata == eighth else 5)
    m = p.mean(axis=0)
    print(f"{name:28s} clear {m[0]:7.3f} ms | rt_render wall {m[1]:8.3f} ms (kernel {m[2]:8.3f}, launch + counters + sync {m[1] - m[2]:6.3f}) | "
          f"all-reduce (1 rank) {m[3]:6.3f} ms | host sync {m[4]:6.3f} ms | {m[5] / m[2] / 1e6:6.2f} G segments/s in the kernel")
    if strata == n_strata:
        whole_kernel, whole_wall = m[2], m[0] + m[1]
    else:
        e_clear, e_wall, e_kernel, e_red = m[0], m[1], m[2], m[3]
serial = frames_per_second(eighth, args.frames, False)
overlap = frames_per_second(eighth, args.frames, True)
whole_frame = frames_per_second(n_strata, 5, True)
print(f"frames of an eighth, back to back: {serial:.3f} ms each waited for, {overlap:.3f} ms with two films in turn; whole frames {whole_frame:.3f} ms")
# prediction for eight GPUs: the slowest rank's frame + an eight-rank all-reduce in place of the one-rank one
ar = args.allreduce_ms
t8_serial = serial - e_red + ar
t8_overlap = max(overlap - min(e_red, overlap - e_wall - e_clear), e_clear + e_wall) + 0.0  # the reduce hides behind the next frame
print(f"PREDICTION (not a measurement), eight GPUs, all-reduce of eight ranks assumed {ar} ms:")
print(f"  every frame waited for : {t8_serial:.3f} ms per frame -> {whole_frame / (8 * t8_serial):.1%} of eight times one GPU")
print(f"  two films in turn      : {t8_overlap:.3f} ms per frame if the reduce hides fully -> {whole_frame / (8 * t8_overlap):.1%}; "
      f"{whole_frame / (8 * (t8_overlap + ar)):.1%} if it does not hide at all")
print(f"  the kernel alone       : {e_kernel:.3f} ms for {eighth} of {n_strata} strata = {whole_kernel / (8 * e_kernel):.1%}")
dist.destroy_process_group()
