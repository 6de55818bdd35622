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


def _runner_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from pcabo import distributed as D
    D.init(backend="gloo")
    import importlib
    er_mod = importlib.import_module("Algorithms.Experiment.ExperimentRunner")
    er = er_mod.ExperimentRunner(algorithms=["pca"], dimensions=[20, 40], problem_ids=list(range(15, 25)), num_runs=30,
                                 root_dir="/tmp", experiment_name="x", progress=False)
    mine = er._my_runs()
    counts = D.gather_best([float(len(mine)), float(sum(sharding.run_cost(r) for r in mine))])
    backend = D.backend_name()
    D.finalize()
    q.put((rank, mine, counts, er._folder("pca"), backend))


def test_sharded_experiment_runner_run_lists_two_ranks_gloo():
    """BASELINE.json configs[3] (30 runs x f15-f24 x d in {20, 40}) through the runner's own partition on 2 ranks: every
    run on exactly one rank, suite order kept inside a rank, balanced cost, rank-suffixed folders, backend recorded."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, 29613, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, c0, f0, b0), (r1, m1, c1, f1, b1) = res
    all_runs = sharding.enumerate_runs(range(15, 25), [20, 40], 30)
    assert sorted(m0 + m1) == sorted(all_runs) and set(m0).isdisjoint(m1)
    assert m0 == [r for r in all_runs if r in set(m0)]                      # suite order inside the rank
    assert c0 == c1 and c0[0][0] + c0[1][0] == 600.0
    assert max(c0[0][1], c0[1][1]) / min(c0[0][1], c0[1][1]) < 1.02
    assert (f0, f1) == ("pca-x-rank0", "pca-x-rank1") and b0 == b1 == "gloo"


def test_nccl_request_without_gpu_exits_instead_of_falling_back(monkeypatch):
    """A rank that cannot use RCCL must stop (non-zero exit), not continue over gloo on its own."""
    import torch
    from pcabo import distributed as D
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit):
        D.init(backend="nccl")


def _runner_worker8(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    import torch
    torch.set_num_threads(1)
    from pcabo import distributed as D
    D.init(backend="gloo")
    import importlib
    er_mod = importlib.import_module("Algorithms.Experiment.ExperimentRunner")
    er = er_mod.ExperimentRunner(algorithms=["pca"], dimensions=[20, 40], problem_ids=list(range(15, 25)), num_runs=30,
                                 root_dir="/tmp", experiment_name="x", progress=False, batched=75, side_by_side=4,
                                 batch_acq_kernel="auto")
    mine = er._my_runs()
    # what a rank would hand to the final gather: one best value per run, padded to the largest share (NaN)
    share = max(len(s) for s in sharding.assign_runs(sharding.enumerate_runs(range(15, 25), [20, 40], 30), world))
    local = [1000.0 * pid + 10.0 * dim + inst for pid, dim, inst in mine] + [float("nan")] * (share - len(mine))
    gathered = D.gather_best(local)
    seen = D.ranks_seen()
    D.finalize()
    q.put((rank, mine, gathered, dict(er.arithmetic_modes), seen))


def test_sharded_experiment_runner_eight_ranks_gloo():
    """BASELINE.json configs[3] on EIGHT ranks (gloo on the CPU): `_my_runs()` partitions the 600 runs, the gather returns
    every run's value exactly once on every rank, and the arithmetic mode `auto` resolves to is the SAME on every rank and
    the same as in a single process - it is taken from the experiment, not from the rank's share (37-38 runs per dimension
    here, 300 in the experiment)."""
    import math
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 8
    procs = [ctx.Process(target=_runner_worker8, args=(r, world, 29617, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    all_runs = sharding.enumerate_runs(range(15, 25), [20, 40], 30)
    mine = [r[1] for r in res]
    assert sorted(x for m in mine for x in m) == sorted(all_runs)
    assert all(r[4] == world for r in res)
    for r in res:
        assert r[2] == res[0][2] or all((a == b) or (math.isnan(a) and math.isnan(b)) for ra, rb in zip(r[2], res[0][2]) for a, b in zip(ra, rb))
    flat = [v for row in res[0][2] for v in row if not math.isnan(v)]
    assert sorted(flat) == sorted(1000.0 * pid + 10.0 * dim + inst for pid, dim, inst in all_runs)
    assert all(r[3] == {20: "device", 40: "device"} for r in res)
    loads = [sum(sharding.run_cost(x) for x in m) for m in mine]
    assert max(loads) / min(loads) < 1.05


def test_arithmetic_mode_is_a_property_of_the_experiment(monkeypatch):
    """`auto` (and an explicit "device") resolve from the experiment's description only: any world size, any rank, any share."""
    import importlib
    er_mod = importlib.import_module("Algorithms.Experiment.ExperimentRunner")
    R = er_mod.resolve_arithmetic_mode
    assert R("auto", 0, 40, 600, 450) == "latency" and R("device", 1, 40, 600, 450) == "latency"      # not batched: the serial classes
    assert R("auto", 30, 40, 30, 450) == "device" and R("auto", 30, 40, 29, 450) == "group"
    assert R("auto", 30, 10, 300, 150) == "group" and R("auto", 30, 20, 300, 250) == "device"
    assert R("auto", 30, 100, 300, 1050) == "group" and R("device", 30, 100, 300, 1050) == "group"     # beyond n <= 512, k <= 40
    assert R("device", 30, 40, 3, 450) == "device" and R("device", 30, 40, 3, 600) == "group"
    assert R("group", 30, 40, 300, 450) == "group" and R("latency", 30, 40, 300, 450) == "latency"
    with pytest.raises(ValueError):
        R("fastest", 30, 40, 300, 450)
    seen = set()
    for world in (1, 2, 8):
        for rank in range(world):
            monkeypatch.setenv("RANK", str(rank)); monkeypatch.setenv("LOCAL_RANK", str(rank)); monkeypatch.setenv("WORLD_SIZE", str(world))
            er = er_mod.ExperimentRunner(algorithms=["pca"], dimensions=[10, 20, 40], problem_ids=[15, 16, 17], num_runs=30,
                                         root_dir="/tmp", experiment_name="x", progress=False, batched=45, batch_acq_kernel="auto")
            seen.add(tuple(sorted(er.arithmetic_modes.items())))
    assert seen == {((10, "group"), (20, "device"), (40, "device"))}
