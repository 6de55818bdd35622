"""Multi-rank plumbing on CPU (gloo, world_size 2): the N > 1 path of bench.py / the sharded runner is
"independent shards + one gather", so this covers the barrier, the max/sum reductions used for timing
and the final best-f gather, plus the run partitioning."""
import os

import pytest
import torch.multiprocessing as mp

from pcabo import sharding


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from pcabo import distributed as D
    r, lr, size = D.init(backend="gloo")
    runs = sharding.enumerate_runs([15, 16], [20, 40], 3)
    mine = sharding.assign_runs(runs, size)[r]
    local_best = [100.0 * r + i for i in range(3)]
    D.barrier()
    mx = D.max_over_ranks(1.0 + r)
    sm = D.sum_over_ranks(float(len(mine)))
    g = D.gather_best(local_best)
    D.finalize()
    q.put((r, mx, sm, g, mine))


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29611, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mx0, sm0, g0, m0), (r1, mx1, sm1, g1, m1) = res
    assert mx0 == mx1 == 2.0
    assert sm0 == sm1 == 12.0                                   # every run assigned exactly once
    assert g0 == g1 == [[0.0, 1.0, 2.0], [100.0, 101.0, 102.0]]
    assert set(m0).isdisjoint(m1) and len(m0) + len(m1) == 12


def test_partition_is_balanced_and_deterministic():
    runs = sharding.enumerate_runs(range(15, 25), [20, 40], 30)       # BASELINE.json configs[3]: 600 runs
    assert len(runs) == 600
    for world in (1, 2, 4, 8):
        shards = sharding.assign_runs(runs, world)
        assert sorted(r for s in shards for r in s) == sorted(runs)
        loads = [sum(sharding.run_cost(r) for r in s) for s in shards]
        assert max(loads) / min(loads) < 1.02
        assert shards == sharding.assign_runs(list(reversed(runs)), world)
    assert sharding.run_settings((15, 40, 7)) == {"budget": 450, "n_doe": 120, "seed": 15407}
    with pytest.raises(ValueError):
        sharding.assign_runs(runs, 0)
