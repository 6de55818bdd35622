"""CPU-side checks of the boundary: the shared library loads and exports every symbol declared in
include/pcabo.h, fails loudly without a device, and the host mirror keeps the reference's surface."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(native):
    header = open(os.path.join(ROOT, "include", "pcabo.h")).read()
    declared = set(re.findall(r"\b(pcabo_[a-z_0-9]+)\s*\(", header)) - {"pcabo_fg_callback"}
    assert declared == set(native.EXPORTS)
    lib = ctypes.CDLL(native.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pcabo_abi_version() == native.ABI_VERSION


def test_no_cpu_fallback_without_device(native):
    if native.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(native.PcaboError) as e:
        native.Context(max_n=32, max_d=4)
    assert e.value.code == -3


def test_batch_and_device_objectives_fail_loudly_without_device(native):
    """The lock-step batch and the device objectives have no CPU path either."""
    if native.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(native.PcaboError) as e:
        native.Batch(2, max_n=32, max_d=4)
    assert e.value.code == -3
    from pcabo.bbob import BBOBProblem
    from pcabo.bbob_device import DeviceObjectives
    with pytest.raises(native.PcaboError):
        DeviceObjectives([BBOBProblem(15, 0, 4), BBOBProblem(21, 1, 4)])
    from pcabo.batchrun import BatchedPCABO
    r = BatchedPCABO([BBOBProblem(15, 0, 4)], [3], 12, 8)
    with pytest.raises(native.PcaboError):
        r.run()                              # DoE runs on the host, the first device call raises
    assert len(r.f_evals[0]) == 8


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "para-ortho-pca-bo_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(base, f)).read()
                assert "pcabo_oracle" not in src and "import oracle" not in src, f


def test_class_surface_matches_reference(native):
    from Algorithms import PCA_BO, AbstractBayesianOptimizer
    assert PCA_BO.TIME_PROFILES == ["SingleTaskGP", "optimize_acqf", "pca"]
    o = PCA_BO(budget=20, n_DoE=5, var_threshold=0.9, acquisition_function="EI", random_seed=7,
               maximization=False, verbose=False, DoE_parameters={"criterion": "center", "iterations": 1000})
    assert isinstance(o, AbstractBayesianOptimizer)
    assert o.acquisition_function_name == "expected_improvement"
    assert (o.budget, o.n_DoE, o.random_seed, o.var_threshold, o.n_components) == (20, 5, 7, 0.9, 0)
    assert o.torch_config["NUM_RESTARTS"] == 10 and o.torch_config["RAW_SAMPLES"] == 512
    assert set(o.timing_logs) == {"SingleTaskGP", "optimize_acqf", "pca"} and o.total_times["pca"] == 0
    assert o.current_best == np.inf and o.number_of_function_evaluations == 0
    with pytest.raises(ValueError, match="Oddly defined name"):
        PCA_BO(budget=10, acquisition_function="thompson")
    with pytest.raises(AssertionError):
        PCA_BO(budget=0)
    with pytest.raises(ValueError):
        o.current_best_index = -1
    o.maximization = True
    assert o.current_best == -np.inf
    with pytest.raises(AttributeError):
        o.acquisition_function = object()
    with pytest.raises(AttributeError):
        o(lambda x: 0.0, 3, None)            # callable problem without bounds
    with pytest.raises(AttributeError):
        o(42, 3, np.array([-1.0, 1.0]))      # neither ioh-like nor callable


def test_bounds_setter_forms():
    from Algorithms import PCA_BO
    from types import SimpleNamespace
    o = PCA_BO(budget=10)
    o.dimension = 3
    o.bounds = np.array([-5.0, 5.0])
    assert o.bounds.shape == (3, 2) and o.bounds[2, 1] == 5.0
    o.bounds = [[-1, 1], [-2, 2], [-3, 3]]
    assert o.bounds[1].tolist() == [-2.0, 2.0]
    o.bounds = SimpleNamespace(lb=np.array([-4.0]), ub=np.array([4.0]))
    assert o.bounds.tolist() == [[-4.0, 4.0]] * 3
    assert o.compute_space_volume() == pytest.approx(512.0)
    with pytest.raises(AttributeError):
        o.bounds = [1.0, 2.0, 3.0]


def test_host_initializers_consume_torch_rng_like_the_oracle():
    import torch
    import pcabo_oracle as O
    from pcabo import initializers as I
    b = np.vstack([-np.arange(1, 7, dtype=float), np.arange(1, 7, dtype=float)])
    torch.manual_seed(11)
    a = I.draw_sobol(b, 64)
    va = np.sin(a).sum(1)
    ia = I.initialize_q_batch(va, 10)
    torch.manual_seed(11)
    c = O.draw_sobol(b, 64)
    ic = O.initialize_q_batch(c, torch.from_numpy(np.sin(c.numpy()).sum(1)), 10)
    assert np.array_equal(a, c.numpy()) and np.array_equal(ia, ic.numpy())
    assert int(np.argmax(va)) in ia


def test_sobol_helpers_equal_the_real_torch_engine_bit_for_bit():
    """pcabo_sobol_scramble + pcabo_sobol_draw (host helpers of libpcabo) against torch.quasirandom.SobolEngine(k, scramble=True):
    same consumption of the CPU generator, same points - including torch's float32 first point - and the same map into a box
    as botorch's draw_sobol_samples (lo + rng * u)."""
    import torch
    from pcabo import initializers as I
    for k in (1, 2, 7, 36, 40, 100):
        lo = np.linspace(-3, 1, k)
        hi = lo + np.linspace(0.5, 7, k)
        for n in (1, 2, 3, 64, 512, 1000):
            torch.manual_seed(5 + k)
            u = torch.quasirandom.SobolEngine(k, scramble=True).draw(n, dtype=torch.float64)
            ref = (torch.from_numpy(lo) + torch.from_numpy(hi - lo) * u).numpy()
            state = torch.get_rng_state()
            torch.manual_seed(5 + k)
            mine = I.draw_sobol(np.vstack([lo, hi]), n)
            assert torch.equal(torch.get_rng_state(), state)
            torch.manual_seed(5 + k)
            assert torch.equal(I.scrambled_sobol_engine(k).draw(n), u) and np.array_equal(mine, ref), (k, n)
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)     # a run's own generator, as in pcabo.batchrun
    a = I.draw_sobol(np.vstack([np.zeros(5), np.ones(5)]), 16, I.scrambled_sobol_engine(5, g1))
    torch.manual_seed(3)
    assert np.array_equal(a, I.draw_sobol(np.vstack([np.zeros(5), np.ones(5)]), 16)) and torch.equal(g1.get_state(), torch.get_rng_state())
    del g2


def test_sobol_rows_call_equals_the_per_run_calls_bit_for_bit():
    """pcabo_sobol_draw_rows (all runs of a lock-step batch in one call, boxes in the packing of pcabo_batch_acq_bounds, points
    straight into the rows the scoring launch packs) against one pcabo_sobol_draw per run; ragged k, a skipped run."""
    import torch
    from pcabo import initializers as I, _native as N
    ks = [3, 40, 1, 17, 36, 40] + [1 + (7 * b) % 40 for b in range(34)]      # (a wide batch)
    kmax, n = 40, 512
    gens = [torch.Generator().manual_seed(70 + b) for b in range(len(ks))]
    engines = [I.scrambled_sobol_engine(k, g) for k, g in zip(ks, gens)]
    engines[2] = None
    boxes = np.full((len(ks), 2 * kmax), np.nan)
    want = []
    for b, k in enumerate(ks):
        lo, hi = np.linspace(-3, 1, k) * (b + 1), np.linspace(1.5, 7, k) + b
        boxes[b, :k], boxes[b, k: 2 * k] = lo, hi
        want.append(None if engines[b] is None else I.draw_sobol(np.vstack([lo, hi]), n, engines[b]))
    out = np.full((len(ks), n * kmax), -7.0)
    views = N.sobol_draw_rows(engines, n, boxes, out)
    for b, k in enumerate(ks):
        if engines[b] is None:
            assert views[b] is None and np.all(out[b] == -7.0)
        else:
            assert views[b].shape == (n, k) and np.array_equal(views[b], want[b]) and np.shares_memory(views[b], out)
            assert np.all(out[b, n * k:] == -7.0)


def test_gc_guard_is_reference_counted_and_undone():
    """pcabo/gcguard.py: freeze + raised young-generation threshold while at least one run is open, everything back to
    what it was afterwards (nested runs: the outermost leave restores)."""
    import gc
    from pcabo import gcguard
    before, frozen_before = gc.get_threshold(), gc.get_freeze_count()
    gcguard.enter()
    assert gc.get_threshold()[0] >= 50000 and gc.get_freeze_count() > 0 and gc.isenabled()
    gcguard.enter()
    gcguard.leave()
    assert gc.get_threshold()[0] >= 50000          # still one run open
    gcguard.leave()
    assert gc.get_threshold() == before and gc.get_freeze_count() == frozen_before
    gcguard.leave()                                 # unbalanced leave is harmless
    assert gc.get_threshold() == before


def test_runner_divides_a_dimension_evenly_over_side_by_side_batches():
    from Algorithms.Experiment.ExperimentRunner import split_evenly
    sizes = lambda n, b, s: [len(p) for p in split_evenly(list(range(n)), b, s)]
    assert sizes(90, 45, 2) == [45, 45] and sizes(30, 30, 2) == [15, 15] and sizes(300, 30, 2) == [30] * 10
    assert sizes(90, 30, 2) == [22, 23, 22, 23] and sizes(119, 30, 2) == [29, 30, 30, 30]      # `batched` is an upper bound
    assert sizes(1, 30, 2) == [1] and sizes(3, 4, 2) == [1, 2] and sizes(6, 2, 1) == [2, 2, 2] and sizes(0, 30, 2) == []
    for n, b, s in [(7, 3, 2), (100, 30, 3), (61, 30, 2), (5, 1, 4), (119, 30, 2), (31, 30, 1)]:
        parts = split_evenly(list(range(n)), b, s)
        assert [x for p in parts for x in p] == list(range(n))              # every run once, in order
        assert max(map(len, parts)) - min(map(len, parts)) <= 1             # even
        assert max(map(len, parts)) <= b                                    # never more runs in a batch than asked for
        assert len(parts) % s == 0 or len(parts) == n                       # whole groups of side-by-side batches


def test_run_side_by_side_drives_every_batch_and_reraises():
    """Host logic of pcabo.batchrun.run_side_by_side with stand-in batches (no device): every batch is started, stepped to
    its budget and finished on its own thread; an exception in one of them surfaces after the others have ended."""
    import threading
    from pcabo.batchrun import run_side_by_side, workers_for

    class Fake:
        def __init__(self, budget, fail_at=None):
            self.budget, self.n, self.fail_at, self.log, self.thread = budget, 0, fail_at, [], None
        def start(self): self.log.append("start"); self.thread = threading.current_thread().name
        def iteration(self):
            if self.fail_at is not None and self.n == self.fail_at:
                raise RuntimeError("boom")
            self.n += 1
        def finish(self): self.log.append("finish")

    a, b = Fake(5), Fake(9)
    run_side_by_side([a, b])
    assert (a.n, b.n) == (5, 9) and a.log == b.log == ["start", "finish"] and a.thread != b.thread
    c, d = Fake(6, fail_at=2), Fake(4)
    with pytest.raises(RuntimeError, match="boom"):
        run_side_by_side([c, d])
    assert d.n == 4 and c.log == d.log == ["start", "finish"]
    one = Fake(3)
    run_side_by_side([one])
    assert one.n == 3 and one.thread == threading.current_thread().name     # a single batch stays on the caller's thread
    assert workers_for(1) == 8 and workers_for(2) == 4 and workers_for(8) == 2


def test_hardware_queue_advice_reads_the_environment_and_never_writes_it(native, monkeypatch):
    """pcabo/_native.py leaves GPU_MAX_HW_QUEUES alone (round 2 set it at import): it only says what is missing."""
    import os
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    msg = native.hw_queues_advice(10)
    assert msg is not None and "GPU_MAX_HW_QUEUES=16" in msg and "unset" in msg
    assert "GPU_MAX_HW_QUEUES" not in os.environ                     # asked, not changed
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert native.hw_queues_advice(8) is None and native.hw_queues_advice(10) is not None
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert native.hw_queues_advice(10) is None
    src = open(os.path.join(ROOT, "para-ortho-pca-bo_amd", "pcabo", "_native.py")).read()
    assert "os.environ.setdefault" not in src and "os.environ[" not in src


def test_host_generator_blob_equals_torchs_generator_on_ten_thousand_picks(native):
    """pcabo/hostrng.py: torch's CPU generator held as its own state blob and advanced by libpcabo's
    host helpers - the Sobol scramble bits (`torch.randint(2, ...)`) and the Boltzmann pick (`torch.multinomial(w, 10)`) of botorch's
    initialize_q_batch.  10 000 random weight vectors on generators at random positions of their streams: the picks AND the
    generator's state after every call are torch's, bit for bit; the rows-at-once form equals row-by-row calls."""
    import torch
    from pcabo import hostrng as H
    from pcabo import initializers as I
    assert H.native_ok()
    rng = np.random.default_rng(2024)
    gens = [(H.HostMT(int(s)), torch.Generator().manual_seed(int(s))) for s in rng.integers(0, 2 ** 31, size=50)]
    for trial in range(10000):
        h, g = gens[trial % len(gens)]
        if trial % 7 == 0:                     # move along the stream by the scramble draws of one engine (k x 30 + k x 30 x 30 bits)
            k = int(rng.integers(1, 6))
            a = np.concatenate([h.randint2((k, 30)).ravel(), h.randint2((k, 30, 30)).ravel()])
            b = torch.cat([torch.randint(2, (k, 30), generator=g).ravel(), torch.randint(2, (k, 30, 30), generator=g).ravel()]).numpy()
            assert np.array_equal(a, b)
        n = int(rng.choice([16, 64, 512]))
        w = np.exp(rng.normal(size=n) * rng.uniform(0.05, 4.0))
        if trial % 11 == 0:
            w[rng.integers(0, n, size=n // 4)] = 0.0                      # zero weights are never picked before positive ones
        got = H.multinomial_rows(w[None], 10, [h], [0])[0]
        want = torch.multinomial(torch.from_numpy(w), 10, generator=g).numpy()
        assert np.array_equal(got, want), trial
        assert np.array_equal(h.blob, g.get_state().numpy()), trial
    # the engine built from a blob equals the one built from torch's generator, and so do the picks of initialize_q_batch_rows
    hs, gs = [H.HostMT(77 + b) for b in range(6)], [torch.Generator().manual_seed(77 + b) for b in range(6)]
    for k in (3, 36):
        for b in range(6):
            e1, e2 = I.scrambled_sobol_engine(k, hs[b]), I.scrambled_sobol_engine(k, gs[b])
            assert np.array_equal(e1.state, e2.state) and np.array_equal(e1.shift, e2.shift)
    vals = rng.normal(size=(6, 512))
    vals[4] = 1.25                                                        # all values tie: the random-permutation path
    with pytest.warns(RuntimeWarning):
        p1 = I.initialize_q_batch_rows(vals, 10, hs, skip={2})
    with pytest.warns(RuntimeWarning):
        p2 = I.initialize_q_batch_rows(vals, 10, gs, skip={2})
    for a, b in zip(p1, p2):
        assert np.array_equal(a, b)
    for b in range(6):
        assert np.array_equal(hs[b].get_state().numpy(), gs[b].get_state().numpy())
        one = I.initialize_q_batch(vals[b] if b != 4 else rng.normal(size=512), 10, generator=hs[b])
        assert len(one) == 10


def test_native_boltzmann_pick_equals_the_torch_path_on_ten_thousand_rows(native):
    """initialize_q_batch for all runs of a batch in one native call (pcabo_boltzmann_pick_rows: Welford statistics, exp, torch's
    multinomial restated, forced arg-max) against the torch path of pcabo.initializers on 10 000 rows of raw-sample scores: same
    picks, same generator states - including rows whose values all tie (random-permutation path) and rows with ties at the top."""
    import warnings
    import torch
    from pcabo import hostrng as H
    from pcabo import initializers as I
    rng = np.random.default_rng(7)
    B = 50
    hs = [H.HostMT(1000 + b) for b in range(B)]
    gs = [torch.Generator().manual_seed(1000 + b) for b in range(B)]
    for rep in range(200):
        vals = rng.normal(size=(B, 512)) * rng.uniform(0.01, 30.0, size=(B, 1)) + rng.normal(size=(B, 1)) * 100.0
        if rep % 10 == 0:
            vals[3] = -7.5                                   # all equal
            vals[5, rng.integers(0, 512, size=200)] = vals[5].max() + 1.0      # many ties at the maximum
            vals[6] = np.round(vals[6])                      # heavy ties everywhere
        skip = {11} if rep % 3 == 0 else set()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = I.initialize_q_batch_rows(vals, 10, hs, skip=skip)
            b = I.initialize_q_batch_rows(vals, 10, gs, skip=skip)
        for r in range(B):
            assert np.array_equal(a[r], b[r]), (rep, r)
    for r in range(B):
        assert np.array_equal(hs[r].get_state().numpy(), gs[r].get_state().numpy()), r


def test_every_lhs_criterion_of_the_reference_surface():
    """pyDOE's four criteria as the reference's LHS_sampler accepts them (AbstractBayesianOptimizer.py:8-103): "center" is pinned by
    the reference's runs (tests/test_reference_kats.py); the other three are restated from the published algorithm - asserted here:
    a Latin hypercube (one point per bin in every column), the candidates' RNG consumption, the criterion no worse than a single draw."""
    from Algorithms.BayesianOptimization.AbstractBayesianOptimizer import LHS_sampler
    from pcabo import lhs as L
    m, d = 12, 4
    for crit in ("center", "maximin", "centermaximin", "correlation", "c", "m", "cm", "corr"):
        np.random.seed(5)
        pts = LHS_sampler(crit, iterations=7)(d, m)
        assert pts.shape == (m, d) and (pts >= 0).all() and (pts < 1).all()
        for j in range(d):
            assert sorted(np.floor(pts[:, j] * m).astype(int)) == list(range(m)), crit          # one point per bin
        after = np.random.rand()
        np.random.seed(5)
        for _ in range(1 if crit in ("center", "c") else 7):                                   # per candidate: rand(m, d) + d permutations
            np.random.rand(m, d)
            for _j in range(d):
                np.random.permutation(m)
        assert np.random.rand() == after, crit
    rs = np.random.RandomState(9)
    one = L._pdist_min(L._lhs_classic(d, m, np.random.RandomState(9)))
    assert L._pdist_min(L.lhs(d, m, "maximin", 25, rs)) >= one
    with pytest.raises(ValueError):
        LHS_sampler("optimal")
    assert np.array_equal(LHS_sampler("center", sample_zero=True)(3, 5)[0], np.zeros(3))


def test_xcd_tile_placement_visits_every_tile_once():
    """`xcd_tile()` (csrc/pcabo_internal.h) re-reads a batched launch's linear work-group id as (run, tile) so that all tiles of a run
    share id % 8 - one XCD under round-robin dispatch.  Restated here: for any grid (gx, gy, B) the map is a bijection onto
    (run, x, y), and within the groups of eight runs every tile of a run has the same id % 8."""
    def xcd_tile(lin, gx, gy, B):
        T = gx * gy
        full = (B & ~7) * T
        if lin < full:
            slot = lin >> 3
            tile = slot % T
            return (slot // T) * 8 + (lin & 7), tile % gx, tile // gx
        z, rem = divmod(lin, T)
        return z, rem % gx, rem // gx
    for gx, gy, B in ((17, 1, 120), (9, 1, 30), (1, 1, 7), (5, 5, 16), (18, 1, 8), (3, 2, 13), (68, 1, 120)):
        seen = {}
        for lin in range(gx * gy * B):
            key = xcd_tile(lin, gx, gy, B)
            assert key not in seen and key[0] < B and key[1] < gx and key[2] < gy
            seen[key] = lin
        assert len(seen) == gx * gy * B
        for run in range(B & ~7):
            assert len({seen[(run, x, y)] % 8 for x in range(gx) for y in range(gy)}) == 1
