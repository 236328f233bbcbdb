"""Batch sharding over the GPUs of one node: every (batch, head, query tile) is independent, so the
attention path partitions by splitting B contiguously - one process per GPU, NO collective on the data path.
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests) is used only for the barriers,
the max-over-ranks of the wall time and, in parity mode, one all_gather of the per-rank outputs.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the batch owned by `rank`; the first batch % world ranks get one extra sample."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(tensors: Sequence[torch.Tensor], world: int, rank: int) -> List[torch.Tensor]:
    """Slice every tensor along dim 0 to this rank's part of the batch (views, no copies)."""
    lo, hi = shard_bounds(tensors[0].shape[0], world, rank)
    return [t[lo:hi] for t in tensors]


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (the bench's wall time); identity when not initialised."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_batch(local_out: torch.Tensor, batch: int) -> torch.Tensor:
    """Parity mode: reassemble the full-batch output from the per-rank shards (uneven shards are padded to the
    largest one for the fixed-size all_gather).  Not used in the timed path."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_out
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]
    big = max(sizes)
    pad = local_out.new_zeros((big,) + tuple(local_out.shape[1:]))
    pad[: sizes[rank]] = local_out
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def run_sharded(fn: Callable[..., torch.Tensor], tensors: Sequence[torch.Tensor], gather: bool = False) -> torch.Tensor:
    """Apply `fn` (e.g. ops.attn_fwd) to this rank's batch shard; optionally gather the full result."""
    if not (dist.is_available() and dist.is_initialized()):
        return fn(*tensors)
    world, rank = dist.get_world_size(), dist.get_rank()
    out = fn(*shard(tensors, world, rank))
    return gather_batch(out, tensors[0].shape[0]) if gather else out


def ranks_seen(device=None) -> int:
    """How many ranks take part in the job, counted by a SUM all-reduce of ones over the process group's transport
    (RCCL over xGMI under "nccl"): the bench reports it so that an N-GPU line is backed by N ranks that really
    answered a collective, not by an environment variable."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_equal(local: torch.Tensor) -> List[torch.Tensor]:
    """all_gather of same-shaped per-rank tensors (weak scaling: every rank holds B_local samples); parity mode only."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local.contiguous())
    return parts


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_ranks(script: str, nproc: int, argv: Sequence[str], env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Start `script` as `nproc` ranks of one node (one process per GPU) through `python -m torch.distributed.run` in a
    CHILD process and return its exit code; the child's stdout / stderr are the caller's.  Must be called before the
    calling process has initialised the GPU (it never does here: the caller only parses arguments) - a process that
    has touched the GPU must not be replaced by, or fork into, GPU work on this pool.  Rendezvous on 127.0.0.1."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script, *argv]
    return subprocess.run(cmd, env=e, timeout=timeout).returncode
