"""One-process-per-GPU plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  The hot path has NO collective: ranks run independent BO runs; the only
exchange is the final gather of best-so-far values (a few hundred bytes - pure latency)."""
from __future__ import annotations

import datetime
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def world() -> tuple:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str = None) -> tuple:
    global _RANKS_SEEN
    rank, local_rank, size = world()
    if size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # PCABO_DIST_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks (ranks then share a device)
            backend = os.environ.get("PCABO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            # RCCL or nothing: a rank that quietly went on over gloo while its peers sit in RCCL hangs the job, and a
            # scaling line must not be able to claim RCCL without having used it.  A rehearsal on a box with fewer GPUs
            # than ranks asks for gloo explicitly (PCABO_DIST_BACKEND=gloo).
            if torch.cuda.device_count() <= local_rank:
                raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); "
                                 "set PCABO_DIST_BACKEND=gloo to rehearse with ranks sharing a device")
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=size,
                                    timeout=datetime.timedelta(seconds=180))
            t = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank))
            dist.all_reduce(t)
            torch.cuda.synchronize()
            if int(t.item()) != size:
                raise SystemExit(f"rank {rank}: RCCL all-reduce saw {t.item()} ranks instead of {size}")
            _RANKS_SEEN = int(t.item())
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=size)
            t = torch.ones(1, dtype=torch.float64)
            dist.all_reduce(t)
            _RANKS_SEEN = int(t.item())
    return rank, local_rank, size


_RANKS_SEEN = 1


def ranks_seen() -> int:
    """Number of ranks the opening all-reduce counted on this process's backend (1 for a single rank)."""
    return _RANKS_SEEN


def backend_name() -> str:
    """Backend the collectives of this process run on ("nccl" = RCCL over xGMI, "gloo", or "none" for a single rank)."""
    return dist.get_backend() if dist.is_initialized() else "none"


def _dev():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_best(local_best: Sequence[float]) -> List[List[float]]:
    """All-gather of each rank's best-so-far values (equal counts per rank padded with NaN by the caller)."""
    if not dist.is_initialized():
        return [list(map(float, local_best))]
    t = torch.tensor(list(local_best), dtype=torch.float64, device=_dev())
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]


def finalize() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()
