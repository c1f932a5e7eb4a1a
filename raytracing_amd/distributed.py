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
import time

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


def spawn_ranks(argv, world, share_gpu=False, extra_env=None, timeout=None):
    """Starts `world` fresh worker processes running `argv` (a command line), one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way torch.distributed.run sets them, waits
    for all of them and returns the worst exit code.  The caller must not have touched the GPU:
    the workers are new processes (fork + exec of the interpreter), the parent only waits.
    Rank 0 inherits stdout, so what it prints is the job's output.  If one worker fails the
    others are ended (by their own pids).  timeout: seconds for the whole job (None: the
    RT_SPAWN_TIMEOUT environment variable, else an hour); ranks that are still running then -- a
    rendezvous or an all-reduce that never completes, a peer stuck in a kernel -- are terminated,
    then killed, and the job fails with exit code 124 instead of holding its GPUs forever."""
    if timeout is None:
        timeout = float(os.environ.get("RT_SPAWN_TIMEOUT", "3600"))
    deadline = time.monotonic() + timeout
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
            if time.monotonic() > deadline:
                print(f"spawn_ranks: ranks {sorted(pending)} still running after {timeout:.0f} s: ending them", file=sys.stderr, flush=True)
                for o in pending:
                    procs[o].terminate()
                grace = time.monotonic() + 10.0
                while time.monotonic() < grace and any(procs[o].poll() is None for o in pending):
                    time.sleep(0.1)
                worst = worst or 124
                break
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
    """Owns one Device and two torch film tensors (3, H, W), bound into it in turn.

    Ordering: the renderer has a torch stream of its own and hands its handle to the device
    backend, so clearing the film, the render kernels and the all-reduce are all issued on (or,
    for RCCL, ordered against) that one stream.  torch's default stream would not do: its handle
    is the null pointer, which the backend reads as "use your own non-blocking stream", and a
    non-blocking stream is not ordered against the null stream.

    Two films: frame k's all-reduce is enqueued without waiting for it (render_async), and frame
    k + 1 is cleared and rendered into the other film meanwhile -- the reduce of a 17 MB film and the
    fixed cost of a launch are what an eighth of a frame per GPU leaves visible.  A film is reused
    by the frame after next, which first waits for the reduce that last read it.

    reduce="torch": torch.distributed's all_reduce (RCCL when the backend is nccl).
    reduce="c-abi": the library's own rt_film_allreduce -- RCCL inside librt_amd.so, the call a
    host without Python makes; the group's id travels through torch.distributed's store once."""

    def __init__(self, device, local_rank=0, reduce="torch"):
        if reduce not in ("torch", "c-abi"):
            raise ValueError("reduce must be 'torch' or 'c-abi'")
        self.dev = device
        self.local_rank = local_rank
        self.reduce_kind = reduce
        self.films = [None, None]
        self.pending = [None, None]  # the reduce in flight on each film
        self.turn = 0
        self.film = None  # the film of the latest frame
        self.staging = None
        self.torch_device = torch.device("cuda", local_rank)
        self.stream = torch.cuda.Stream(device=self.torch_device)
        if self.stream.cuda_stream == 0:
            raise RuntimeError("torch returned the null stream for a new torch.cuda.Stream")
        self.dev.set_stream(self.stream.cuda_stream)
        self._comm_ready = False

    def bind_film(self, width, height):
        self.wait()
        with torch.cuda.stream(self.stream):
            self.films = [torch.zeros((3, height, width), dtype=torch.float32, device=self.torch_device) for _ in range(2)]
        self.stream.synchronize()
        self.turn = 0
        self.film = self.films[0]
        self.dev.film_bind(self.film.data_ptr(), width, height)
        self.staging = None
        return self.film

    def _join_c_abi_group(self):
        """One group per process group: rank 0 makes the id (rt_comm_unique_id), the others read it."""
        from . import api
        if self._comm_ready:
            return
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
            box = [api.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, device=self.torch_device if dist.get_backend() == "nccl" else None)
            uid = box[0]
        else:
            rank, world, uid = 0, 1, api.comm_unique_id()
        self.dev.comm_init_rank(uid, rank, world)
        self._comm_ready = True

    def _wait_for(self, k):
        work, self.pending[k] = self.pending[k], None
        if work is not None:
            work.wait()  # the current stream waits for the reduce; the host does not

    def wait(self):
        """Both films complete on the host's side: every reduce enqueued so far has finished."""
        with torch.cuda.stream(self.stream):
            for k in range(2):
                self._wait_for(k)
        self.stream.synchronize()

    def render_async(self, cam, n_strata, rank, world, mode, seed=1, reduce=True, **render_args):
        """Renders this rank's strata of every pixel into the film that is not being reduced and
        enqueues that film's all-reduce; returns (film, Stats) when the render has finished, the reduce
        possibly not: the film is complete after wait() -- or after the next call but one, which
        takes the film back."""
        if self.films[0] is None or tuple(self.films[0].shape[1:]) != (cam.image_height, cam.image_width):
            self.bind_film(cam.image_width, cam.image_height)
        k = self.turn
        self.turn ^= 1
        film = self.films[k]
        begin, end = partition_strata(n_strata, world, rank)
        stats = None
        with torch.cuda.stream(self.stream):
            self._wait_for(k)  # the reduce that last read this film
            film.zero_()
            self.dev.film_bind(film.data_ptr(), cam.image_width, cam.image_height)
            if end > begin:
                stats = self.dev.render(cam, mode=mode, seed=seed, stratum_begin=begin, stratum_end=end, **render_args)
            if reduce:
                if self.reduce_kind == "c-abi":
                    self._join_c_abi_group()
                    self.dev.film_allreduce()  # on the handle's stream = self.stream, behind the render
                elif dist.is_available() and dist.is_initialized():
                    if dist.get_backend() == "gloo":
                        if self.staging is None:
                            self.staging = torch.empty(film.shape, dtype=torch.float32, device="cpu", pin_memory=True)
                        reduce_film(film, staging=self.staging)
                    else:  # RCCL, on its own stream behind this one; the next frame does not wait for it
                        self.pending[k] = dist.all_reduce(film, op=dist.ReduceOp.SUM, async_op=True)
        self.film = film
        return film, stats

    def render(self, cam, n_strata, rank, world, mode, seed=1, reduce=True, **render_args):
        """Renders this rank's strata of every pixel and reduces; returns (film sums, Stats).
        The film is complete (on every rank) when this returns."""
        film, stats = self.render_async(cam, n_strata, rank, world, mode, seed=seed, reduce=reduce, **render_args)
        self.wait()
        return film, stats
