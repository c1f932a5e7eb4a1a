"""The N > 1 path on the GPU: ranks started the way `python bench.py --gpus N` starts them,
rendering through DistributedRenderer.  A one-GPU box has one card, so the ranks share GPU 0 and
reduce over gloo (RCCL refuses two ranks on one device); everything else -- stratum split,
stream ordering of clear / render / reduce over consecutive frames -- is the multi-GPU code."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("scene,world,backend,reduce", [("cornell_book3", 2, "gloo", "torch"), ("cornell_main", 3, "gloo", "torch"),
                                                        ("cornell_book3", 1, "nccl", "torch"), ("cornell_book3", 1, "nccl", "c-abi")])
def test_ranks_sharing_gpu0_reproduce_the_one_rank_render(tmp_path, scene, world, backend, reduce):
    """(The last two cases are one rank with a process group over RCCL: clear, render and the all-reduce
    of the film through the backend the 8-GPU job uses -- torch.distributed's all_reduce, enqueued
    without waiting while the next frame renders into the other film, and the library's own
    rt_film_allreduce with the group's id handed over through the process group.)"""
    import raytracing_amd as ra
    from raytracing_amd import distributed as rdist
    seeds = [5, 6, 7]
    width, spp = 96, 25  # 25 strata do not divide evenly over 2 or 3 ranks
    rc = rdist.spawn_ranks([sys.executable, str(ROOT / "tests" / "dist_worker.py"), str(tmp_path), scene, str(width), str(spp),
                            ",".join(map(str, seeds)), backend, reduce], world, share_gpu=True, timeout=500)
    assert rc == 0
    sb = ra.SceneBuilder()
    cam = getattr(ra.scenes, scene)(sb, width=width, spp=spp)
    dev = ra.Device(0)
    try:
        dev.upload(sb)
        renderer = rdist.DistributedRenderer(dev, 0)
        mode = ra.RT_MODE_REF_EXACT if scene == "cornell_main" else ra.RT_MODE_BOOK3
        for seed in seeds:
            film, st = renderer.render(cam.build(), cam.effective_spp, 0, 1, mode, seed=seed)
            want = film.cpu().numpy()
            assert want.sum() > 0
            films = [np.load(tmp_path / f"film_r{r}_s{seed}.npy") for r in range(world)]
            for f in films[1:]:
                assert np.array_equal(f, films[0])  # the all-reduce leaves the same sums everywhere
            # same samples, summed in another order (fp32 atomics, then the reduce)
            assert np.allclose(films[0], want, rtol=2e-5, atol=1e-5), float(np.abs(films[0] - want).max())
            seg = np.load(tmp_path / f"seg_s{seed}.npy")
            assert int(seg[0]) == st.segments and int(seg[1]) == st.camera_rays == width * width * cam.effective_spp
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the plain command starts two ranks itself and
    rank 0 prints the one JSON line (here as a rehearsal, both ranks on GPU 0 over gloo)."""
    import json
    import os
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--share-gpu",
                        "--backend", "gloo", "--spp", "16"], capture_output=True, text=True, timeout=800, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and "rehearsal" in out
    # both ranks' segments are in the total: 16 strata of the full frame
    assert out["config"]["spp"] == 16 and out["config"]["segments_per_camera_ray"] > 1.0
    assert abs(out["config"]["segments_per_step"] / (1200 * 1200 * 16) - out["config"]["segments_per_camera_ray"]) < 1e-2
