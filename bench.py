#!/usr/bin/env python3
"""Benchmark of the render hot path: ray segments/s on the Book-3 Cornell box.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one full frame of the workload: every rank renders the whole 1200x1200 image for
its own contiguous share of the 961 (= floor(sqrt(1000))^2, camera.rs:341-342) strata, then the
fp32 film sums are all-reduced (RCCL).  The default pipeline is the persistent path kernel (one
launch per step); --wavefront takes the ray-queue pipeline (scan/generate/extend/shade).  Total work is fixed, so scaling is "strong"; the value
is (segments traced by all ranks) / (slowest rank's time), in Mrays/s.  A segment is one
closest-hit query issued by the integrator (one world.hit at camera.rs:444).
Scene and camera are synthetic (generated in code, seeded); inputs are resident in HBM before
the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# Importing the package loads the HIP library but makes no HIP call: a process that goes on to
# start the ranks of a multi-GPU run (below) has not touched the GPU when it does.
import raytracing_amd as ra  # noqa: E402
from raytracing_amd import distributed as rdist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, spec
VALU_PEAK_TFLOPS = 157.3   # fp32 vector peak (packed); 78.6 with scalar v_fma_f32

WORKLOADS = {
    # name: (scene function, kwargs, estimator mode, description)
    "c3": (ra.scenes.cornell_book3, dict(width=1200, spp=1000), ra.RT_MODE_BOOK3,
           "Book-3 Cornell box, glass sphere + light in the light list, 1200x1200, 961 spp, depth 50, mixture PDF"),
    "c1": (ra.scenes.cornell_main, dict(width=600, spp=20), ra.RT_MODE_REF_EXACT,
           "main.rs Cornell box as shipped, 600x600, 16 spp, depth 50, light-only sampling"),
    "c2": (ra.scenes.book1_final, dict(width=1200, spp=500), ra.RT_MODE_BOOK3,
           "Book-1 final scene, ~488 spheres under a BVH, defocus blur, 1200x675, 484 spp"),
    "c4": (ra.scenes.book2_final, dict(width=800, spp=10000), ra.RT_MODE_BOOK3,
           "Book-2 final scene, 800x800, 10000 spp"),
    "c5": (ra.scenes.stress, dict(n_prims=1_000_000), ra.RT_MODE_BOOK3, "1M random spheres/quads, 3840x2160, 4096 spp"),
}

# ray-queue bytes one segment moves through each wavefront kernel (16-byte records, rt_kernels.h)
EXTEND_BYTES = 32 + 16           # read origin/time + direction/origin-ref, write the hit record
SHADE_READ_BYTES = 32 + 24 + 16  # ray, throughput/pixel, stratum/bounce, hit record
SHADE_WRITE_BYTES = 56           # a surviving ray's records
FILM_BYTES_PER_PATH = 24         # three fp32 read-modify-writes when a path deposits radiance
# algorithmic flops per segment measured by the oracle's counters (flops_per_segment below) on
# the full workloads; used when the CPU leg is skipped, recomputed when it runs
# (c2, c4, c5: full scene content, a few strata of the full frame, the reference's own binary BVH
# walked left child first -- 40.9, 12.2 and 340.8 box tests per segment).  Averaged over the
# segments that carry weight, the ones the device traces (see live_segments).
FLOPS_PER_SEGMENT = {"c3": 559.1, "c1": 681.4, "c2": 1310.7, "c4": 1162.7, "c5": 9528.7}
# Counter digests of the dominant kernel per workload, made from rocprofv3 PMC passes of this
# binary by tools/pmc_to_bench.py (counters need runs of their own: FETCH_SIZE and WRITE_SIZE do
# not fit one pass).  Each entry names the summary under profiles/ it was made from.
PMC_DIGEST = ROOT / "bench_pmc.json"


def pmc_digest(workload, kernel):
    """{"hbm_bytes_per_segment", "lanes_active", "valu_busy", "source"} for this workload's kernel, or None."""
    try:
        entry = json.loads(PMC_DIGEST.read_text()).get(workload)
    except (OSError, ValueError):
        return None
    return entry if entry and entry.get("kernel", "") == kernel else None


def live_segments(cnt):
    """The oracle's segments that carry weight.  It traces a path whose weight has become exactly 0
    to its end like the reference does and counts those segments apart; the device stops such a
    path.  Rates and per-segment figures of both sides are over this population."""
    return cnt["segments"] - cnt["zero_weight_segments"]


def cpu_baseline(scene_fn, kw, mode, n_strata_sample, threads, shrink=1):
    """The oracle (C++ f64 restatement of the reference) on a bounded sample of the workload:
    the full frame for `n_strata_sample` of the strata -- at 1/shrink of the resolution per axis
    where the full frame would take minutes (same scene, same camera, same path statistics).  Test
    infrastructure used as a checker and as the reported baseline only."""
    from oracle import config_oracle
    from oracle import oracle_py as orc
    osb = orc.scene()
    kw = dict(kw)
    if scene_fn in (ra.scenes.cornell_main, ra.scenes.cornell_book3):
        kw["toml_objects"] = config_oracle.toml_objects
    cam = scene_fn(osb, **kw)
    if shrink > 1:
        cam.image_width, cam.image_height = max(2, cam.image_width // shrink), max(2, cam.image_height // shrink)
    t0 = time.perf_counter()
    _, _, cnt = orc.render(osb, cam, mode=mode, seed=1, stratum_begin=0, stratum_end=n_strata_sample, threads=threads)
    dt = time.perf_counter() - t0
    return cnt, dt, cam


def flops_per_segment(cnt):
    """Algorithmic work of one segment from the oracle's counters: 24/box test, 30/sphere test,
    +30/sphere hit record, 13/quad plane test, +37/quad interior test, 150 for shading."""
    s = max(live_segments(cnt), 1)  # the oracle counts the tests of these segments only
    return (24 * cnt["box_tests"] + 30 * cnt["sphere_tests"] + 30 * cnt["sphere_accepts"] + 13 * cnt["quad_plane_tests"] +
            37 * cnt["quad_interior_tests"]) / s + 150.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (changes the workload)")
    ap.add_argument("--pool", type=int, default=0, help="rays in flight per GPU (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-strata", type=int, default=6, help="strata of the full frame the CPU baseline renders")
    ap.add_argument("--wavefront", action="store_true", help="ray-queue pipeline instead of the persistent path kernel")
    ap.add_argument("--regroup", choices=["auto", "on", "off"], default="auto",
                    help="persistent kernel on a flat world: paths regrouped per wave by the step they wait for, or lane-bound")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="process-group backend (default: nccl = RCCL)")
    ap.add_argument("--reduce", default="torch", choices=["torch", "c-abi"],
                    help="the film all-reduce: torch.distributed's (RCCL under the nccl backend), or the library's own rt_film_allreduce "
                         "(RCCL inside librt_amd.so, what a host without Python calls)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU machine: every rank uses GPU 0 (needs --backend gloo); the line is marked")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.share_gpu and (args.backend or "gloo") != "gloo":
        raise SystemExit("--share-gpu needs --backend gloo: RCCL refuses two ranks on one GPU")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: this process becomes the launcher.  It has made no
        # HIP or torch.cuda call (device_count() does not initialise the GPU); the N ranks are fresh
        # processes, rank 0 prints the JSON line to the inherited stdout, this one only waits.
        have = torch.cuda.device_count()
        if have < 1:
            raise SystemExit("bench.py needs a GPU: the product has no CPU path")
        if have < args.gpus and not args.share_gpu:
            raise SystemExit(f"--gpus {args.gpus} but only {have} GPU(s) are visible (use --share-gpu --backend gloo to rehearse on one)")
        raise SystemExit(rdist.spawn_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus,
                                           share_gpu=args.share_gpu))
    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
        args.backend = "gloo"
    rank, world, local = rdist.init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)

    scene_fn, kw, mode, desc = WORKLOADS[args.workload]
    kw = dict(kw)
    if args.spp:
        kw["spp"] = args.spp
    sb = ra.SceneBuilder()
    cam = scene_fn(sb, **kw)
    dev = ra.Device(local)
    dev.upload(sb)
    rcam = cam.build()
    n_strata = cam.effective_spp
    renderer = rdist.DistributedRenderer(dev, local, reduce=args.reduce)
    renderer.bind_film(cam.image_width, cam.image_height)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    extra = dict(wavefront=True, pool_size=args.pool, time_every=8) if args.wavefront else {}
    if args.regroup != "auto":
        extra["regroup"] = args.regroup == "on"
    last_note = [time.perf_counter()]

    def note(what):
        # frames of the large workloads take a minute each: say so on stderr now and then, so that
        # a harness watching for output does not take the run for hung
        now = time.perf_counter()
        if rank == 0 and now - last_note[0] > 20.0:
            print(f"[bench] {what}", file=sys.stderr, flush=True)
            last_note[0] = now

    for w in range(args.warmup):
        renderer.render(rcam, n_strata, rank, world, mode, seed=1, **extra)
        note(f"warmup step {w + 1} of {args.warmup} done")
    barrier()
    seg = 0
    kernel_ms = np.zeros(5)
    launches = np.zeros(5)
    timed = np.zeros(5)
    cam_rays = 0
    regrouped = False
    t0 = time.perf_counter()
    for step in range(args.steps):
        # kernels are bracketed by HIP events on the stream they run on: the one launch of the
        # persistent kernel, or every 8th wavefront iteration (rotating from step to step; an
        # event is a barrier packet on the stream, timing all of them costs ~10 %)
        # (two films in turn: a frame's all-reduce is in flight while the next frame is rendered)
        _, st = renderer.render_async(rcam, n_strata, rank, world, mode, seed=1 + step, time_kernels=True, **extra)
        if st is not None:
            seg += st.segments
            cam_rays += st.camera_rays
            regrouped = st.regrouped
            kernel_ms += np.array(st.kernel_ms)
            timed += np.array(st.timed_launches)
            launches += np.array(st.launches)
        note(f"step {step + 1} of {args.steps} done")
    renderer.wait()  # the last frames' reduces
    barrier()
    elapsed = time.perf_counter() - t0

    on_host = world > 1 and dist.get_backend() == "gloo"
    stats = torch.tensor([elapsed, float(seg), float(cam_rays)], dtype=torch.float64, device="cpu" if on_host else "cuda")
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_seg, total_cam = float(stats[1]), float(stats[2])
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = total_seg / elapsed / 1e6
    # dominant kernel of this rank, timed with HIP events on the stream the kernels run on
    names = ["generate", "extend", "shade", "scan", "paths"]
    per_launch_ms = kernel_ms / np.maximum(timed, 1)  # average launch duration, from the timed launches
    k = int(np.argmax(per_launch_ms * launches))
    survive = 1.0 - cam_rays / max(seg, 1)  # every path's last segment does not continue
    bytes_per_seg = {"extend": EXTEND_BYTES, "shade": SHADE_READ_BYTES + survive * SHADE_WRITE_BYTES,
                     "paths": FILM_BYTES_PER_PATH * cam_rays / max(seg, 1)}.get(names[k], 56.0)
    avg_launch_ms = per_launch_ms[k]
    seg_per_launch = seg / max(launches[k], 1)
    achieved_gbs = bytes_per_seg * seg_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    fps = FLOPS_PER_SEGMENT.get(args.workload)
    out = {
        "metric": "ray segments/s", "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        **({"rehearsal": "all ranks share GPU 0; not a scaling measurement"} if args.share_gpu else {}),
        "config": {"workload": desc, "name": args.workload, "image": [cam.image_width, cam.image_height],
                   "spp": n_strata, "max_depth": cam.max_depth, "estimator": "book3" if mode == ra.RT_MODE_BOOK3 else "ref-exact",
                   "pipeline": "wavefront" if args.wavefront else ("persistent, paths regrouped per wave" if regrouped else "persistent"),
                   "parallelism": f"spp-split x{world}, one fp32 film all-reduce per step ({'rt_film_allreduce, C ABI' if args.reduce == 'c-abi' else 'torch.distributed'}), "
                                  "two films in turn",
                   "segments_per_step": int(total_seg / args.steps), "segments_per_camera_ray": round(total_seg / max(total_cam, 1), 3)},
        "time_to_1000spp_s": round(ms_per_step * 1e-3 * 1000.0 / n_strata, 4),
        "kernel_ms_per_step": {n: round(float(per_launch_ms[i] * launches[i]) / args.steps, 3) for i, n in enumerate(names)},
    }
    # ray-queue / film traffic of the dominant kernel against HBM (the secondary roofline)
    # the regrouping variant of the persistent kernel is k_paths_regroup on a flat world and
    # k_paths_pool behind a tree
    flat_world = dev.accel_info()["kind"] == "flat"
    paths_kernel = ("k_paths_regroup" if flat_world else "k_paths_pool") if regrouped else "k_paths"
    hbm = {"bound": "hbm", "kernel": paths_kernel if names[k] == "paths" else "k_" + names[k], "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": None,
           "bytes_per_segment": round(bytes_per_seg, 1), "segments_per_launch": int(seg_per_launch),
           "avg_launch_ms": round(float(avg_launch_ms), 4), "launches": int(launches[k]), "timed_launches": int(timed[k])}
    if world == 1 and not args.no_cpu_baseline:
        # a bounded sample: strata of the full frame, the frame shrunk where a stratum alone takes minutes
        cpu_strata, shrink = {"c3": (args.cpu_strata, 1), "c1": (16, 1), "c2": (4, 1), "c4": (2, 1), "c5": (1, 8)}[args.workload]
        frame = "full frame" if shrink == 1 else f"frame at 1/{shrink} of the resolution per axis"
        cnt, dt, _ = cpu_baseline(scene_fn, kw, mode, cpu_strata, threads=1, shrink=shrink)
        fps = flops_per_segment(cnt)
        out["cpu_baseline"] = {"value": round(live_segments(cnt) / dt / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port",
                               "sample": f"{frame}, {cpu_strata} of {n_strata} strata ({cnt['camera_rays']} camera rays, "
                                         f"{live_segments(cnt)} segments that carry weight -- the population the device traces -- of "
                                         f"{cnt['segments']} the reference's loop traces, {dt:.1f} s); C++ f64 restatement of the "
                                         "reference, not the Rust binary"}
        nthreads = min(16, os.cpu_count() or 1)
        many = max(cpu_strata, 8) if args.workload in ("c3", "c1") else cpu_strata * 4
        cnt2, dt2, _ = cpu_baseline(scene_fn, kw, mode, min(many, n_strata), threads=nthreads, shrink=shrink)
        out["cpu_baseline_all_cores"] = {"value": round(live_segments(cnt2) / dt2 / 1e6, 3), "unit": "Mrays/s", "cores": nthreads,
                                         "kind": "port", "sample": f"{frame}, {min(many, n_strata)} strata, OpenMP rows, {dt2:.1f} s"}
        out["time_to_1000spp_cpu_1core_s"] = round(dt / cpu_strata * 1000.0 * shrink * shrink, 1)
    # The path has no dense contraction and moves almost nothing through HBM: its roofline is
    # the fp32 vector peak.  achieved = algorithmic flops per segment (reference arithmetic,
    # FMA = 2, counted by the oracle) x segments per launch / the kernel's launch duration.
    if fps:
        kernel_name = paths_kernel if names[k] == "paths" else "k_" + names[k]
        tflops = fps * seg_per_launch / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        if names[k] != "paths":
            # a segment's arithmetic is spread over the four wavefront kernels: price them together
            kernel_name = "k_scan+k_generate+k_extend+k_shade"
            all_ms = float(np.sum(per_launch_ms * launches))
            tflops = fps * seg / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0
        pmc = pmc_digest(args.workload, kernel_name) if names[k] == "paths" else None
        out["roofline"] = {"bound": "valu", "kernel": kernel_name, "achieved": round(tflops, 3), "peak": VALU_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(tflops / VALU_PEAK_TFLOPS, 5),
                           "traffic": round(pmc["hbm_bytes_per_segment"] * seg_per_launch) if pmc else None,
                           "traffic_unit": "bytes per launch, HBM side: (2 x FETCH_SIZE + WRITE_SIZE) per segment of the counter "
                                           "passes x segments of this launch",
                           **({"lanes_active": pmc["lanes_active"], "valu_busy": pmc["valu_busy"], "pmc_source": pmc["source"]} if pmc else {}),
                           "flops_per_segment": round(fps, 1), "segments_per_launch": int(seg_per_launch),
                           "avg_launch_ms": round(float(avg_launch_ms), 4), "launches": int(launches[k]),
                           "timed_launches": int(timed[k]),
                           "note": "fp32 vector peak with packed FMA; 78.6 TFLOP/s with scalar v_fma_f32"}
        out["roofline_hbm"] = hbm
    else:
        out["roofline"] = hbm
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
