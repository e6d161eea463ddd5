"""Batch-sharded multi-GPU inference (no data-path collective) and data-parallel training (one gradient all-reduce per
step, bucketed and overlapped with the backward pass): one process per GPU.

The reference shards its sample list with accelerate's `PartialState().split_between_processes`
(examples/brushnet/test_brushnet.py:163-168): a static contiguous split where the first `n % world` ranks get
one extra item and nothing is communicated.  `shard_range` reproduces that partition; the only
collectives used by the harness are the barrier and the max-over-ranks of the elapsed time that bench.py's
contract asks for (RCCL when the tensors are on the GPU, gloo on the CPU for tests).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple, TypeVar

import torch
import torch.distributed as dist

T = TypeVar("T")


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(backend: str = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank_world()
    # a torchrun environment initialises the group even for one rank, so a single-GPU box exercises the same RCCL
    # rendezvous / barrier / all-reduce calls the multi-GPU runs make
    if (world > 1 or "TORCHELASTIC_RUN_ID" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def torchrun_argv(nproc: int) -> List[str]:
    """The `python -m torch.distributed.run` prefix for `nproc` ranks on this node WITHOUT a caller-chosen port.

    A port found by bind(0) / close and then handed to `--master-port` can be taken by another process in between
    (EADDRINUSE on a busy box).  The c10d rendezvous on endpoint port 0 lets the agent's TCPStore bind a free port itself and
    keep it open; the workers re-use that store (MASTER_PORT is the store's port), so there is no window.  127.0.0.1 twice:
    the container hostname may not resolve."""
    import sys
    import uuid
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
            "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", f"--rdzv-id={uuid.uuid4().hex}", "--local-addr=127.0.0.1"]


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of this rank's contiguous share (accelerate.PartialState.split_between_processes)."""
    per, extra = divmod(n_items, world)
    start = rank * per + min(rank, extra)
    return start, start + per + (1 if rank < extra else 0)


def shard(items: Sequence[T], rank: int, world: int) -> List[T]:
    a, b = shard_range(len(items), rank, world)
    return list(items[a:b])


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def tuned_once(warm: Callable[[], None], save: Optional[Callable[[], None]] = None, reload: Optional[Callable[[], None]] = None) -> None:
    """Warm-up with a possibly cold GEMM tune cache on several ranks: rank 0 runs `warm` ALONE (it autotunes every new
    (shape -> tile, split-K) once and persists the winners: hip.tune_save), the other ranks wait at a barrier, re-read the cache
    (hip.tune_reload) and only then run `warm` themselves — one tuning pass per node instead of one per rank (an 8-rank start used
    to time every candidate tile eight times, on eight GPUs contending for the host).  Ranks whose shapes differ from rank 0's (a
    ragged last batch shard) still tune their own misses.  A single process just runs `warm`.

    `warm` MUST BE COLLECTIVE-FREE (no gradient exchange, no max_over_ranks): rank 0 runs it while the other ranks sit in a
    collective (the barrier), so a collective inside it would pair up with that barrier and hang or corrupt the job.  If rank 0
    fails, the barrier is still reached and EVERY rank raises (the others would otherwise wait for the watchdog)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        warm()
        return
    if save is None or reload is None:
        from . import hip
        save, reload = save or hip.tune_save, reload or hip.tune_reload
    err: Optional[BaseException] = None
    if dist.get_rank() == 0:
        try:
            warm()
            save()
        except BaseException as e:          # noqa: BLE001 — re-raised below, after the other ranks have been told
            err = e
    flag = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64,
                        device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)          # doubles as the barrier the waiting ranks sit in
    if err is not None:
        raise err
    if float(flag.item()) > 0.0:
        raise RuntimeError("tuned_once: rank 0 failed during its warm-up / autotune pass (see its traceback)")
    if dist.get_rank() != 0:
        reload()
        warm()
    dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class GradBuckets:
    """The gradient exchange of data-parallel training (accelerate's DDP wrap, train_brushnet_mirror.py:1267-1269; the
    all-reduce happens inside `accelerator.backward`, :1459): every rank holds the full model, gradients are AVERAGED.

    Built for the flat gradient arenas and for xGMI: the arena is cut into fixed buckets (default 64 Mi floats = 256 MiB:
    large enough that RCCL's ring / mesh step time is bandwidth- not latency-bound on 153 GB/s links, small enough that
    several are in flight under the backward pass).  The tape reports every finished parameter gradient; when all
    parameters that touch a bucket are final the bucket is all-reduced asynchronously on a side stream (ordered after the
    compute stream by an event) while backward keeps running.  finish() flushes what is left, waits, and divides by the
    world size.  With one rank everything is a no-op."""

    def __init__(self, models, bucket_floats: int = 64 * 1024 * 1024):
        self.models = [m for m in models if m.flat_g is not None]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # MF_FORCE_GRAD_SYNC=1: run the bucketed exchange even with one rank (exercises the RCCL path on a 1-GPU box)
        self.force = os.environ.get("MF_FORCE_GRAD_SYNC") == "1" and dist.is_initialized()
        self.bucket_floats = int(bucket_floats)
        self._stream = None
        # training.GraphedTrainStep sets this while it captures: a bucket that becomes ready is reported to it (it cuts the graph
        # there and issues the exchange between two graph segments at replay) instead of being all-reduced during the capture
        self.capture_hook = None
        self._plan = []
        for m in self.models:
            n = m.num_arena_floats()
            nb = max(1, -(-n // self.bucket_floats))
            pending = [0] * nb
            where = {}
            for name, (a, shape, _meta) in m._pmap.items():
                numel = int(torch.Size(shape).numel())
                bs = range(a // self.bucket_floats, (a + max(numel, 1) - 1) // self.bucket_floats + 1)
                where[name] = list(bs)
                for b in bs:
                    pending[b] += 1
            self._plan.append(dict(model=m, n=n, nb=nb, pending0=pending, where=where))
        self._state = None

    def begin(self, tape) -> None:
        self._state = [dict(pending=list(p["pending0"]), sent=[False] * p["nb"], handles=[]) for p in self._plan]
        self._by_grad = {}
        for pi, p in enumerate(self._plan):
            self._by_grad[p["model"].flat_g.untyped_storage().data_ptr()] = pi
        # MF_GRAD_SYNC_SERIAL=1 (tools/run_8gpu.sh): no exchange under the backward pass — finish() sends every bucket afterwards; the
        # step-time difference against the default is what the overlap hides
        serial = os.environ.get("MF_GRAD_SYNC_SERIAL") == "1"
        tape.on_param_grad = self._on_param if ((self.world > 1 or self.force) and not serial) else None

    def _on_param(self, param) -> None:
        pi = self._by_grad.get(param.grad.untyped_storage().data_ptr())
        if pi is None:
            return
        p, st = self._plan[pi], self._state[pi]
        # a bucket's countdown assumes ONE "gradient final" signal per parameter and step.  A parameter used by two tape ops
        # (weight sharing, a module called twice, recomputation) would be signalled twice and release its bucket while a later
        # contribution is still being accumulated into it: refuse that loudly instead of all-reducing a partial gradient
        done = st.setdefault("done", set())
        if param.name in done:
            raise RuntimeError(f"GradBuckets: the gradient of {param.name!r} was reported final twice in one step (a parameter "
                               "shared by several operators): per-use counting is not built — keep such parameters out of the "
                               "bucketed exchange")
        done.add(param.name)
        for b in p["where"].get(param.name, ()):
            st["pending"][b] -= 1
            if st["pending"][b] == 0 and not st["sent"][b]:
                self._send(pi, b)

    def _send(self, pi: int, b: int) -> None:
        p, st = self._plan[pi], self._state[pi]
        st["sent"][b] = True
        if self.capture_hook is not None:
            self.capture_hook(pi, b)
            return
        self.send_bucket(pi, b)

    def send_bucket(self, pi: int, b: int) -> None:
        """All-reduce bucket b of model pi asynchronously on the side stream, ordered after everything the current stream holds."""
        p, st = self._plan[pi], self._state[pi]
        lo, hi = b * self.bucket_floats, min((b + 1) * self.bucket_floats, p["n"])
        buf = p["model"].flat_g[lo:hi]
        if buf.is_cuda and dist.get_backend() == "gloo":
            # test path (ranks sharing one GPU, no RCCL): stage the bucket through the host
            torch.cuda.current_stream(buf.device).synchronize()
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        elif buf.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=buf.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            self._stream.wait_event(ev)
            with torch.cuda.stream(self._stream):
                st["handles"].append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
        else:
            st["handles"].append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> None:
        if self.world == 1 and not self.force:
            return
        self.flush()
        self.wait()
        self.scale()

    def flush(self) -> None:
        """Send every bucket the backward pass has not released (parameters that received no gradient this step)."""
        for pi, p in enumerate(self._plan):
            for b in range(p["nb"]):
                if not self._state[pi]["sent"][b]:
                    self._send(pi, b)

    def wait(self) -> None:
        """Order the current stream after every exchange in flight."""
        for pi, p in enumerate(self._plan):
            for h in self._state[pi]["handles"]:
                h.wait()
            self._state[pi]["handles"] = []
            g = p["model"].flat_g
            if g.is_cuda and self._stream is not None:
                torch.cuda.current_stream(g.device).wait_stream(self._stream)

    def scale(self) -> None:
        """Sum -> mean over the ranks (one streaming kernel per arena)."""
        for p in self._plan:
            g = p["model"].flat_g[: p["n"]]
            if g.is_cuda:
                from . import hip
                hip.axpby_n([g], [1.0 / self.world], out=g)
            else:
                g.mul_(1.0 / self.world)          # CPU (gloo) test path only


    def time_all_reduce(self, reps: int = 3) -> Optional[float]:
        """Milliseconds one un-overlapped all-reduce of every gradient bucket takes (outside any training step: the arenas are
        summed `reps` times and left scaled — call it after the last optimizer step; BASELINE.json configs[3] asks for this
        number beside samples/s).  None with one rank."""
        if self.world == 1 and not self.force:
            return None
        import time
        bufs = []
        for p in self._plan:
            for b in range(p["nb"]):
                bufs.append(p["model"].flat_g[b * self.bucket_floats: min((b + 1) * self.bucket_floats, p["n"])])
        if not bufs[0].is_cuda or dist.get_backend() == "gloo":
            return None
        torch.cuda.synchronize()
        for buf in bufs:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)           # warm the communicator for these sizes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for buf in bufs:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3


def gather_mean(value: torch.Tensor) -> float:
    """accelerator.gather(loss.repeat(bs)).mean() (train_brushnet_mirror.py:1454-1457): the logging loss over all ranks."""
    v = value.detach().float().reshape(1).clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        if v.is_cuda or dist.get_backend() != "nccl":
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            v /= dist.get_world_size()
    return float(v.item())
