"""The N>1 path on CPU: two gloo ranks render disjoint stratum ranges (with the oracle standing
in for the device, since this machine has no GPU), all-reduce the film sums, and must
reproduce the one-rank render of all strata."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import raytracing_amd as ra
    from raytracing_amd import distributed as rdist
    from oracle import oracle_py as orc
    import helpers
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    _, osb, cam = helpers.build_both(ra.scenes.cornell_book3, width=24, spp=16)
    begin, end = rdist.partition_strata(cam.effective_spp, world, rank)
    film, _, cnt = orc.render(osb, cam, mode=orc.MODE_BOOK3, seed=3, stratum_begin=begin, stratum_end=end)
    t = torch.from_numpy(film.astype(np.float32)).permute(2, 0, 1).contiguous()  # planar, like the device film
    rdist.reduce_film(t)
    seg = torch.tensor([float(cnt["segments"])], dtype=torch.float64)
    dist.all_reduce(seg)
    np.save(Path(out_dir) / f"film_{rank}.npy", t.numpy())
    np.save(Path(out_dir) / f"seg_{rank}.npy", seg.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_strata_split_matches_single_rank(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, str(ROOT / "tests"))
    import raytracing_amd as ra
    from oracle import oracle_py as orc
    import helpers
    _, osb, cam = helpers.build_both(ra.scenes.cornell_book3, width=24, spp=16)
    film, _, cnt = orc.render(osb, cam, mode=orc.MODE_BOOK3, seed=3)
    want = film.astype(np.float32).transpose(2, 0, 1)
    f0, f1 = np.load(tmp_path / "film_0.npy"), np.load(tmp_path / "film_1.npy")
    assert np.array_equal(f0, f1)  # all-reduce leaves the same sums on both ranks
    assert np.allclose(f0, want, rtol=1e-5, atol=1e-6)  # same samples, different fp32 summation order
    assert float(np.load(tmp_path / "seg_0.npy")[0]) == cnt["segments"]


def test_reduce_film_is_a_noop_without_a_group():
    from raytracing_amd import distributed as rdist
    t = torch.ones(3, 2, 2)
    assert rdist.reduce_film(t) is t and float(t.sum()) == 12.0
    assert rdist.init_from_env() == (0, 1, 0)
