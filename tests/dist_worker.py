"""One rank of the multi-rank GPU test (started by tests/test_gpu_distributed.py through
spawn_ranks): renders its share of the strata of a few consecutive frames through
DistributedRenderer, all ranks on GPU 0 with the gloo backend, and saves every reduced film.
With backend "nccl" (one rank only: RCCL refuses two ranks on one device) the film reduce goes
through RCCL on the renderer's stream."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main(out_dir, scene, width, spp, seeds, backend="gloo", reduce="torch"):
    import torch
    import torch.distributed as dist
    import raytracing_amd as ra
    from raytracing_amd import distributed as rdist

    rank, world, local = rdist.init_from_env(backend=backend, single_rank_group=backend == "nccl")
    assert dist.is_initialized() and dist.get_backend() == backend
    assert local == 0
    torch.cuda.set_device(0)
    sb = ra.SceneBuilder()
    cam = getattr(ra.scenes, scene)(sb, width=width, spp=spp)
    dev = ra.Device(0)
    dev.upload(sb)
    renderer = rdist.DistributedRenderer(dev, 0, reduce=reduce)
    rcam = cam.build()
    mode = ra.RT_MODE_REF_EXACT if scene == "cornell_main" else ra.RT_MODE_BOOK3
    # consecutive frames into two films in turn, none waited for before the next is rendered: clear, render
    # and reduce of every film must stay ordered, and a film must not be cleared under its own reduce
    frames = [renderer.render_async(rcam, cam.effective_spp, rank, world, mode, seed=seed) for seed in seeds[:2]]
    renderer.wait()
    kept = [(f.clone(), st) for f, st in frames]
    for seed in seeds[2:]:
        film, st = renderer.render(rcam, cam.effective_spp, rank, world, mode, seed=seed)
        kept.append((film.clone(), st))
    for seed, (film, st) in zip(seeds, kept):
        seg = torch.tensor([float(st.segments if st else 0), float(st.camera_rays if st else 0)], dtype=torch.float64,
                           device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(seg)
        seg = seg.cpu()
        np.save(Path(out_dir) / f"film_r{rank}_s{seed}.npy", film.cpu().numpy())
        if rank == 0:
            np.save(Path(out_dir) / f"seg_s{seed}.npy", seg.numpy())
    dist.barrier()
    dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), [int(s) for s in sys.argv[5].split(",")],
         *(sys.argv[6:8]))
