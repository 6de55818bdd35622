"""Run-level sharding across the GPUs of one node.

A "run" is one (problem_id, dimension, instance) tuple - a closed computation with its own seed
(reference: Algorithms/Experiment/ExperimentRunner.py:137-146).  Runs never exchange data, so the
multi-GPU path is: enumerate runs, balance them by estimated cost, give each rank (one process per
GPU) its list, and gather the best-so-far values once at the end.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

Run = Tuple[int, int, int]   # (problem_id, dimension, instance)


def enumerate_runs(problem_ids: Sequence[int], dimensions: Sequence[int], num_runs: int) -> List[Run]:
    """Same nesting order as ioh's BBOB suite iteration used by the reference runner."""
    return [(fid, dim, inst) for fid in problem_ids for dim in dimensions for inst in range(num_runs)]


def run_settings(run: Run, budget_factor: int = 10, doe_factor: float = 3.0):
    """budget / n_DoE / seed of a run exactly as ExperimentRunner.py:144-146 derives them."""
    fid, dim, inst = run
    return {"budget": budget_factor * dim + 50, "n_doe": int(doe_factor * dim),
            "seed": 1000 * fid + 10 * dim + inst}


def run_cost(run: Run, budget_factor: int = 10, doe_factor: float = 3.0) -> float:
    """Relative cost ~ sum over BO iterations of n^2 (Gram/Cholesky/acquisition all scale ~n^2..n^3)."""
    s = run_settings(run, budget_factor, doe_factor)
    return float(sum(n * n for n in range(s["n_doe"], s["budget"])))


def assign_runs(runs: Sequence[Run], world_size: int, budget_factor: int = 10, doe_factor: float = 3.0) -> List[List[Run]]:
    """Longest-processing-time-first greedy partition; deterministic (ties by run tuple)."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = sorted(runs, key=lambda r: (-run_cost(r, budget_factor, doe_factor), r))
    loads = [0.0] * world_size
    shards: List[List[Run]] = [[] for _ in range(world_size)]
    for r in order:
        i = min(range(world_size), key=lambda j: (loads[j], j))
        shards[i].append(r)
        loads[i] += run_cost(r, budget_factor, doe_factor)
    return shards
