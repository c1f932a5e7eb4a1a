"""Multi-GPU rendering: one process per GPU, strata split across ranks, one film reduce.

Every rank renders the FULL frame for a disjoint, contiguous range of stratum indices
(stratum = sj * sqrt_spp + si of camera.rs:390-391).  The random stream of a sample is keyed
by (pixel, stratum), so the set of samples does not depend on how many ranks share the work.
The only exchange is one sum all-reduce of the fp32 film (W*H*3 floats) at the end: RCCL over
xGMI when the process group's backend is "nccl", gloo on CPU tensors in tests.
"""
import os

import torch
import torch.distributed as dist


def partition_strata(n_strata: int, world_size: int, rank: int):
    """Contiguous [begin, end) for `rank`; the first n_strata % world_size ranks get one more."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, rem = divmod(n_strata, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torch.distributed.run sets them."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local


def reduce_film(film: torch.Tensor, group=None):
    """In-place sum of the per-rank film sums; a no-op on one rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
    return film


class DistributedRenderer:
    """Owns one Device and a torch film tensor (3, H, W) bound into it."""

    def __init__(self, device, local_rank=0):
        self.dev = device
        self.local_rank = local_rank
        self.film = None

    def bind_film(self, width, height):
        self.film = torch.zeros((3, height, width), dtype=torch.float32, device=torch.device("cuda", self.local_rank))
        self.dev.set_stream(torch.cuda.current_stream(self.film.device).cuda_stream)
        self.dev.film_bind(self.film.data_ptr(), width, height)
        return self.film

    def render(self, cam, n_strata, rank, world, mode, seed=1, reduce=True, **render_args):
        """Renders this rank's strata of every pixel and reduces; returns (film sums, Stats)."""
        if self.film is None or tuple(self.film.shape[1:]) != (cam.image_height, cam.image_width):
            self.bind_film(cam.image_width, cam.image_height)
        self.film.zero_()
        begin, end = partition_strata(n_strata, world, rank)
        stats = None
        if end > begin:
            stats = self.dev.render(cam, mode=mode, seed=seed, stratum_begin=begin, stratum_end=end, **render_args)
        if reduce:
            reduce_film(self.film)
        return self.film, stats
