"""Multi-GPU rendering: one process per GPU, strata split across ranks, one film reduce.

Every rank renders the FULL frame for a disjoint, contiguous range of stratum indices
(stratum = sj * sqrt_spp + si of camera.rs:390-391).  The random stream of a sample is keyed
by (pixel, stratum), so the set of samples does not depend on how many ranks share the work.
The only exchange is one sum all-reduce of the fp32 film (W*H*3 floats) at the end: RCCL over
xGMI when the process group's backend is "nccl"; with "gloo" (tests, rehearsals of several
ranks on one GPU) the film is staged through pinned host memory.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def partition_strata(n_strata: int, world_size: int, rank: int):
    """Contiguous [begin, end) for `rank`; the first n_strata % world_size ranks get one more."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, rem = divmod(n_strata, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def init_from_env(backend=None, single_rank_group=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torch.distributed.run sets them.
    single_rank_group: make a process group for one rank too (the film reduce then goes through
    the backend as it does with more ranks: a rehearsal of the RCCL path on a one-GPU machine)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
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


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv, world, share_gpu=False, extra_env=None):
    """Starts `world` fresh worker processes running `argv` (a command line), one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way torch.distributed.run sets them, waits
    for all of them and returns the worst exit code.  The caller must not have touched the GPU:
    the workers are new processes (fork + exec of the interpreter), the parent only waits.
    Rank 0 inherits stdout, so what it prints is the job's output.  If one worker fails the
    others are ended (by their own pids)."""
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK="0" if share_gpu else str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this driver
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if rank == 0 else subprocess.DEVNULL))
    worst = 0
    pending = set(range(world))
    try:
        while pending:
            for r in sorted(pending):
                try:
                    rc = procs[r].wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if rc != 0:
                    worst = worst or rc
                    for o in pending:  # a rank died: its peers would wait for it forever
                        procs[o].terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def reduce_film(film: torch.Tensor, group=None, staging=None):
    """In-place sum of the per-rank film sums; a no-op without a process group (a group of one
    rank still calls the backend).  A device film under the gloo backend goes through `staging` (a
    host tensor of the same shape, pinned for speed) or a temporary."""
    if not (dist.is_available() and dist.is_initialized()):
        return film
    if film.is_cuda and dist.get_backend(group) == "gloo":
        host = staging if staging is not None else torch.empty(film.shape, dtype=film.dtype, device="cpu")
        host.copy_(film)  # blocking copy on the current stream
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        film.copy_(host)
        return film
    dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
    return film


class DistributedRenderer:
    """Owns one Device and a torch film tensor (3, H, W) bound into it.

    Ordering: the renderer has a torch stream of its own and hands its handle to the device
    backend, so clearing the film, the render kernels and the all-reduce are all issued on (or,
    for RCCL, ordered against) that one stream.  torch's default stream would not do: its handle
    is the null pointer, which the backend reads as "use your own non-blocking stream", and a
    non-blocking stream is not ordered against the null stream."""

    def __init__(self, device, local_rank=0):
        self.dev = device
        self.local_rank = local_rank
        self.film = None
        self.staging = None
        self.torch_device = torch.device("cuda", local_rank)
        self.stream = torch.cuda.Stream(device=self.torch_device)
        if self.stream.cuda_stream == 0:
            raise RuntimeError("torch returned the null stream for a new torch.cuda.Stream")
        self.dev.set_stream(self.stream.cuda_stream)

    def bind_film(self, width, height):
        with torch.cuda.stream(self.stream):
            self.film = torch.zeros((3, height, width), dtype=torch.float32, device=self.torch_device)
        self.stream.synchronize()
        self.dev.film_bind(self.film.data_ptr(), width, height)
        self.staging = None
        return self.film

    def render(self, cam, n_strata, rank, world, mode, seed=1, reduce=True, **render_args):
        """Renders this rank's strata of every pixel and reduces; returns (film sums, Stats).
        The film is complete (on every rank) when this returns."""
        if self.film is None or tuple(self.film.shape[1:]) != (cam.image_height, cam.image_width):
            self.bind_film(cam.image_width, cam.image_height)
        begin, end = partition_strata(n_strata, world, rank)
        stats = None
        with torch.cuda.stream(self.stream):
            self.film.zero_()
            if end > begin:
                stats = self.dev.render(cam, mode=mode, seed=seed, stratum_begin=begin, stratum_end=end, **render_args)
            if reduce:
                if (self.staging is None and dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo"):
                    self.staging = torch.empty(self.film.shape, dtype=torch.float32, device="cpu", pin_memory=True)
                reduce_film(self.film, staging=self.staging)
        self.stream.synchronize()
        return self.film, stats
