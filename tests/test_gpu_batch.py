"""Batched contexts: B runs advancing in lock-step through pcabo_batch_* must each take, bit for bit,
the path the same run takes alone in `Algorithms.PCA_BO` (same kernels, same per-run grids; per-run RandomState /
torch.Generator seeded like the reference's global generators)."""
import numpy as np
import pytest
import torch

from pcabo.bbob import BBOBProblem

pytestmark = pytest.mark.gpu


def _single(fid, inst, dim, budget, n_doe, seed, acq_kernel="latency"):
    from Algorithms import PCA_BO
    opt = PCA_BO(budget=budget, n_DoE=n_doe, random_seed=seed, maximization=False, acq_kernel=acq_kernel)
    opt(BBOBProblem(fid, inst, dim))
    return np.vstack(opt.x_evals), np.array(opt.f_evals), opt.current_best, opt.current_best_index


def _batched(fid, insts, dim, budget, n_doe, seeds, acq_kernel="group"):
    from pcabo.batchrun import BatchedPCABO
    r = BatchedPCABO([BBOBProblem(fid, i, dim) for i in insts], seeds, budget, n_doe, acq_kernel=acq_kernel)
    r.run()
    return r


@pytest.mark.parametrize("mode", ["group", "slab"])
@pytest.mark.parametrize("dim,budget,n_doe,B", [(10, 70, 30, 5), (40, 200, 120, 3)])
def test_batched_runs_equal_single_runs_bit_for_bit(native, dim, budget, n_doe, B, mode):
    """mode "group": the batch's default - L-BFGS-B rounds through the throughput kernel k_acq_group; the single run takes
    the same kernel (acq_kernel="group").  mode "slab": the per-query kernels on both sides (the single run's default,
    resident kernel included: same arithmetic as one launch per evaluation)."""
    torch.set_num_threads(4)
    insts = list(range(B))
    seeds = [15000 + 10 * dim + i for i in insts]
    r = _batched(15, insts, dim, budget, n_doe, seeds, "group" if mode == "group" else "latency")
    for b, i in enumerate(insts):
        X, f, best, bi = _single(15, i, dim, budget, n_doe, seeds[b], "group" if mode == "group" else "latency")
        assert np.array_equal(np.vstack(r.x_evals[b]), X), (dim, b)
        assert np.array_equal(np.array(r.f_evals[b]), f), (dim, b)
        assert r.current_best[b] == best and r.current_best_index[b] == bi


def test_batch_state_matches_single_context_calls(native):
    """One lock-step conditioning of 4 runs (different data, same n and d): every run's GP state, search box, scores and
    one optimize call equal what a stand-alone context computes from the same inputs, bit for bit."""
    rng = np.random.default_rng(21)
    B, n, d, q = 4, 150, 12, 512
    X = rng.uniform(-5, 5, (B, n, d))
    y = rng.normal(size=(B, n)) * 50 + 300
    ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
    noise = rng.normal(0, 1e-8, (B, n, d))
    bt = native.Batch(B, max_n=200, max_d=d, max_q=q)
    bt.wpca_gp_condition_begin(X, ranks, noise, y)
    res = bt.wpca_results()
    boxes = bt.acq_bounds()
    raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * rng.uniform(size=(q, res[b]["k"])) for b in range(B)]
    best = [float(y[b].min()) for b in range(B)]
    vals, status = bt.gp_wait_eval(raw, best)
    assert not status.any()
    ics = [raw[b][:10] for b in range(B)]
    outs, status = bt.optimize_acqf(ics, boxes, best)
    assert not status.any()
    z = [outs[b][0][int(np.argmax(outs[b][1]))] for b in range(B)]
    xs = bt.inverse_map(z)
    assert len({r["k"] for r in res}) >= 1
    for b in range(B):
        c = native.Context(max_n=200, max_d=d, max_q=q)
        c.set_option(native.OPT_GROUP_ACQ, 1)             # the batch's default kernel for value+gradient evaluations
        r1 = c.wpca_gp_condition(X[b], y[b], ranks=ranks[b], noise=noise[b])
        assert r1["k"] == res[b]["k"]
        for key in ("data_mean", "pca_mean", "components", "evr"):
            assert np.array_equal(r1[key], res[b][key]), (b, key)
        assert np.array_equal(c.acq_bounds(), boxes[b])
        v1 = c.gp_wait_eval(raw[b], best[b])
        assert np.array_equal(v1, vals[b]), b
        st1, stb = c.gp_state(), bt.ctx[b].gp_state()
        for key in ("L", "R", "alpha", "norm_bounds"):
            assert np.array_equal(st1[key], stb[key]), (b, key)
        assert np.array_equal(c.gram(), bt.ctx[b].gram())
        cand, v, info, failed = c.optimize_acqf(ics[b], boxes[b], best[b])
        assert np.array_equal(cand, outs[b][0]) and np.array_equal(v, outs[b][1]) and np.array_equal(info, outs[b][2])
        assert failed == outs[b][3]
        assert np.array_equal(c.inverse_map(z[b]), xs[b])
        # the single-context calls also work on a batch's member
        v2, g2 = bt.ctx[b].acq_eval(ics[b], best[b])
        v3, g3 = c.acq_eval(ics[b], best[b])
        assert np.array_equal(v2, v3) and np.array_equal(g2, g3)
        c.close()
    bt.close()


def test_device_objectives_match_the_host_restatements(native):
    """BBOB f15-f24 on the device (csrc/kernels_bbob.hip) against pcabo.bbob on the same points, all ten functions at
    d = 5 / 20 / 40, instances 0..2: values to 1e-11 relative (other summation order), the out-of-box rule exactly."""
    from pcabo.bbob import FUNCTIONS
    from pcabo.bbob_device import DeviceObjectives
    rng = np.random.default_rng(8)
    for d in (5, 20, 40):
        probs = [BBOBProblem(fid, inst, d) for fid in sorted(FUNCTIONS) for inst in range(3)]
        dev = DeviceObjectives(probs)
        for rep in range(4):
            X = rng.uniform(-5, 5, (len(probs), d))
            if rep == 1:
                X[::4, 0] = 5.0 + 1e-9                      # just outside: penalised, not evaluated
            if rep == 2:
                X = np.stack([p.optimum.x + 1e-3 * rng.normal(size=d) for p in probs]).clip(-5, 5)
            f, raw, oob = dev.evaluate(X)
            for b, p in enumerate(probs):
                outside = bool(np.any(X[b] < -5.0) or np.any(X[b] > 5.0))
                assert bool(oob[b]) == outside
                if outside:
                    assert f[b] == 1000.0
                else:
                    want = p.raw(X[b])
                    assert abs(raw[b] - want) <= 1e-11 * max(1.0, abs(want)), (p.meta_data.problem_id, d, raw[b], want)
                    assert abs(f[b] - (want + p.f_opt)) <= 1e-11 * max(1.0, abs(want + p.f_opt))
        dev.close()


def test_batched_run_with_device_objectives(native):
    """The whole lock-step loop with the objectives on the device: f15/f16/f17 (BASELINE.json configs[2]) at d = 10, one
    batch.  Against the same batch with host objectives: same DoE, same first BO candidates; trajectories may part later
    where a 1e-13 difference in f flips a rank (chaos, EXPERIMENTS.md section 6), so the check is on the first iterations."""
    from pcabo.batchrun import BatchedPCABO
    runs = [(fid, inst) for fid in (15, 16, 17) for inst in (0, 1)]
    out = []
    for dev in (False, True):
        r = BatchedPCABO([BBOBProblem(f, i, 10) for f, i in runs], [1000 * f + 100 + i for f, i in runs], 45, 30,
                         device_objective=dev)
        r.run()
        out.append(r)
    for b in range(len(runs)):
        xa, xb = np.vstack(out[0].x_evals[b]), np.vstack(out[1].x_evals[b])
        assert np.array_equal(xa[:31], xb[:31])                       # DoE + first candidate: same inputs, same device path
        fa, fb = np.array(out[0].f_evals[b]), np.array(out[1].f_evals[b])
        assert np.abs(fa[:31] - fb[:31]).max() <= 1e-10 * max(1.0, np.abs(fa[:31]).max())
        assert len(out[1].problems[b].log) == out[1].problems[b].evaluations


def test_experiment_runner_batched_writes_the_same_files(native, tmp_path):
    """ExperimentRunner(batched=B): the PCA_BO runs of a dimension advance in lock-step; with the per-query kernels on both
    sides the files must equal those of the run-by-run runner byte for byte (apart from the wall-time attributes)."""
    import os
    from Algorithms import ExperimentRunner
    from pcabo import iohlog
    outs = []
    for batched in (0, 4):
        root = tmp_path / f"b{batched}"
        er = ExperimentRunner(algorithms=["pca"], dimensions=[5], problem_ids=[15, 20], num_runs=3, budget_factor=5,
                              doe_factor=2.0, root_dir=str(root), experiment_name="experiment", progress=False, batched=batched,
                              batch_acq_kernel="latency")
        er.run_experiment()
        assert len(er.results) == 6
        outs.append((root, sorted((r["problem_id"], r["instance"], r["best"]) for r in er.results)))
    assert outs[0][1] == outs[1][1]
    for fid, name in ((15, "RastriginRotated"), (20, "Schwefel")):
        rel = os.path.join("pca-experiment", f"data_f{fid}_{name}", f"IOHprofiler_f{fid}_DIM5.dat")
        a, b = open(os.path.join(outs[0][0], rel)).read(), open(os.path.join(outs[1][0], rel)).read()
        assert a == b


def test_runs_on_concurrent_host_threads_in_one_process(native):
    """BASELINE.json configs[3] "one run per stream": several PCA_BO runs inside ONE process, a host thread and a context
    (= a HIP stream) each.  The library notices the other contexts and serves every evaluation with a launch of its own
    (resident kernels of several runs cannot share the chip); each run must reproduce what it produces alone, bit for bit.
    (numpy's and torch's GLOBAL generators are shared by the threads of a process, so the reference's own classes cannot run
    concurrently with RNG parity; the runs here use the lock-step driver's per-run generators, one batch of one run each.)"""
    import threading
    from pcabo.batchrun import BatchedPCABO

    def one(inst, out):
        r = BatchedPCABO([BBOBProblem(15, inst, 10)], [15100 + inst], 60, 30)
        r.run()
        out[inst] = (np.vstack(r.x_evals[0]), np.array(r.f_evals[0]))

    alone = {}
    for i in range(3):
        one(i, alone)
    together = {}
    threads = [threading.Thread(target=one, args=(i, together)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(3):
        assert np.array_equal(alone[i][0], together[i][0]) and np.array_equal(alone[i][1], together[i][1]), i


def test_a_parked_run_does_not_disturb_the_others(native):
    """A run that cannot go on (the reference's own dynamics can blow its search box up until botorch meets a NaN
    gradient and raises) is parked: the other runs of the batch must still take, bit for bit, the path they take alone."""
    from pcabo.batchrun import BatchedPCABO
    insts = [0, 1, 2]
    seeds = [15100 + i for i in insts]
    r = BatchedPCABO([BBOBProblem(15, i, 10) for i in insts], seeds, 60, 30)
    r.start()
    for it in range(30):
        if it == 7:
            r._park(1, r.n, "parked by the test")
        r.iteration()
    r.finish()
    assert r.failed[1] == (37, "parked by the test") and len(r.f_evals[1]) == 37
    assert r.failed[0] is None and r.failed[2] is None
    for b in (0, 2):
        X, f, best, bi = _single(15, insts[b], 10, 60, 30, seeds[b], "group")
        assert np.array_equal(np.vstack(r.x_evals[b]), X) and np.array_equal(np.array(r.f_evals[b]), f)


def test_batches_side_by_side_equal_one_batch(native, tmp_path):
    """run_side_by_side: the runs of a cell as two lock-step batches on two host threads (gang workers re-sized through
    pcabo_batch_set_workers) - every run must come out bit for bit as in ONE batch of all of them; the experiment runner with
    side_by_side=2 must write the files it writes with side_by_side=1."""
    import os
    from pcabo.batchrun import BatchedPCABO, run_side_by_side, workers_for
    from Algorithms import ExperimentRunner
    dim, budget, n_doe, insts = 10, 70, 30, list(range(6))
    seeds = [15000 + 10 * dim + i for i in insts]
    whole = _batched(15, insts, dim, budget, n_doe, seeds)
    halves = [BatchedPCABO([BBOBProblem(15, i, dim) for i in insts[t::2]], seeds[t::2], budget, n_doe, workers=workers_for(2))
              for t in range(2)]
    run_side_by_side(halves)
    for t in range(2):
        for j, i in enumerate(insts[t::2]):
            assert np.array_equal(np.vstack(halves[t].x_evals[j]), np.vstack(whole.x_evals[i])), i
            assert np.array_equal(np.array(halves[t].f_evals[j]), np.array(whole.f_evals[i])), i
    outs = []
    for sbs in (1, 2):
        root = tmp_path / f"s{sbs}"
        er = ExperimentRunner(algorithms=["pca"], dimensions=[5, 10], problem_ids=[15], num_runs=4, budget_factor=5,
                              doe_factor=2.0, root_dir=str(root), experiment_name="experiment", progress=False, batched=2,
                              side_by_side=sbs)
        er.run_experiment()
        assert len(er.results) == 8 and not er.failed_runs
        outs.append(root)
    for d in (5, 10):
        rel = os.path.join("pca-experiment", "data_f15_RastriginRotated", f"IOHprofiler_f15_DIM{d}.dat")
        assert open(os.path.join(outs[0], rel)).read() == open(os.path.join(outs[1], rel)).read()


def test_batch_worker_count_can_change_between_calls(native):
    """pcabo_batch_set_workers between two iterations of a batch: same candidates as without the change."""
    from pcabo.batchrun import BatchedPCABO
    dim, budget, n_doe, insts = 10, 50, 30, [0, 1, 2, 3, 4]
    seeds = [15000 + 10 * dim + i for i in insts]
    ref = _batched(15, insts, dim, budget, n_doe, seeds)
    r = BatchedPCABO([BBOBProblem(15, i, dim) for i in insts], seeds, budget, n_doe)
    r.start()
    try:
        for it, w in zip(range(budget - n_doe), [0, 1, 3, 0, 5, 2] * 10):
            if w:
                r._batch.set_workers(w)
            r.iteration()
    finally:
        r.finish()
    for b in range(len(insts)):
        assert np.array_equal(np.vstack(r.x_evals[b]), np.vstack(ref.x_evals[b])), b
    with pytest.raises(Exception):
        native.Batch(2, max_n=40, max_d=5).set_workers(0)


def test_rccl_gather_best_through_the_c_abi(native):
    """pcabo_comm_* / pcabo_gather_best (include/pcabo.h): the final all-gather of best-so-far values over RCCL without
    torch.distributed.  One GPU here, so one rank: the communicator comes up, the gather returns the rank's own values, twice
    (a larger payload re-sizes the device buffer), and argument errors are reported."""
    uid = native.comm_unique_id()
    assert len(uid) == 128
    comm = native.Comm(uid, world=1, rank=0, device=0)
    try:
        a = np.array([3.5, -1.25, 1e300])
        assert np.array_equal(comm.gather_best(a), a.reshape(1, 3))
        b = np.linspace(-5, 5, 300)
        assert np.array_equal(comm.gather_best(b), b.reshape(1, 300))
    finally:
        comm.close()
    with pytest.raises(native.PcaboError):
        native.Comm(uid, world=2, rank=5, device=0)


def test_strided_inputs_and_call_halves(native):
    """pcabo_batch_set_input_strides: X / y handed over as the first n rows of [B][budget][d] / [B][budget] arrays give the state
    the dense arrays give, bit for bit; the begin / end halves of the waiting calls give what the blocking calls give, and an _end
    without its _begin is refused (PCABO_ERR_ARG) instead of waiting for nothing."""
    rng = np.random.default_rng(8)
    B, n, budget, d, q = 3, 90, 140, 9, 512
    Xfull = rng.uniform(-5, 5, (B, budget, d))
    yfull = rng.normal(size=(B, budget)) * 40 + 100
    ranks = np.argsort(np.argsort(yfull[:, :n], axis=1), axis=1) + 1
    noise = rng.normal(0, 1e-8, (B, n, d))
    outs = []
    for strided in (False, True):
        bt = native.Batch(B, max_n=budget, max_d=d, max_q=q, device_lbfgsb=1)
        X = Xfull[:, :n] if strided else np.ascontiguousarray(Xfull[:, :n])
        y = yfull[:, :n] if strided else np.ascontiguousarray(yfull[:, :n])
        bt.wpca_gp_condition_begin(X, ranks, noise, y)
        res = bt.wpca_results()
        boxes = bt.acq_bounds()
        raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * np.random.default_rng(b).uniform(size=(q, res[b]["k"])) for b in range(B)]
        best = [float(y[b].min()) for b in range(B)]
        if strided:                       # the halves
            with pytest.raises(native.PcaboError):
                bt.gp_eval_end((np.zeros((B, q * d)), q, np.zeros(B), 0, native.ACQ_LOG_EI))
            tok = bt.gp_eval_begin(raw, best)
            while bt.busy():
                pass
            vals, status = bt.gp_eval_end(tok)
            with pytest.raises(native.PcaboError):
                bt.optimize_end((10, 5))
            tok = bt.optimize_begin([raw[b][:10] for b in range(B)], boxes, best)
            assert tok is not None
            o, st = bt.optimize_end(tok)
            z = [o[b][0][int(np.argmax(o[b][1]))] for b in range(B)]
            with pytest.raises(native.PcaboError):
                bt.inverse_map_end()
            bt.inverse_map_begin(z)
            x = bt.inverse_map_end()
        else:
            vals, status = bt.gp_wait_eval(raw, best)
            o, st = bt.optimize_acqf([raw[b][:10] for b in range(B)], boxes, best)
            z = [o[b][0][int(np.argmax(o[b][1]))] for b in range(B)]
            x = bt.inverse_map(z)
        assert not status.any() and not st.any()
        outs.append((res, boxes, vals, o, x, [bt.ctx[b].gp_state() for b in range(B)]))
        bt.close()
    (r0, b0, v0, o0, x0, s0), (r1, b1, v1, o1, x1, s1) = outs
    assert np.array_equal(v0, v1) and np.array_equal(x0, x1)
    for b in range(B):
        assert r0[b]["k"] == r1[b]["k"] and np.array_equal(r0[b]["components"], r1[b]["components"])
        assert np.array_equal(b0[b], b1[b])
        for key in ("L", "R", "alpha"):
            assert np.array_equal(s0[b][key], s1[b][key]), (b, key)
        assert np.array_equal(o0[b][0], o1[b][0]) and np.array_equal(o0[b][1], o1[b][1]) and np.array_equal(o0[b][2], o1[b][2])


def _single_vanilla(fid, inst, dim, budget, n_doe, seed):
    from Algorithms import Vanilla_BO
    opt = Vanilla_BO(budget=budget, n_DoE=n_doe, random_seed=seed, maximization=False)
    opt(BBOBProblem(fid, inst, dim))
    return np.vstack(opt.x_evals), np.array(opt.f_evals)


@pytest.mark.parametrize("fid,dim,budget,n_doe,B", [(15, 10, 70, 30, 4), (20, 5, 75, 10, 5)])
def test_batched_vanilla_runs_equal_single_runs_bit_for_bit(native, fid, dim, budget, n_doe, B):
    """BatchedVanillaBO (pcabo_batch_gp_condition_begin: rows D-H of B runs on the raw points, no PCA) against the reference-surface
    class Algorithms.Vanilla_BO run by run: same DoE, same raw samples and picks from the run's own generator, same candidates -
    bit for bit with the per-query kernels on both sides; and the device-resident optimiser against its host-stepped twin."""
    from pcabo.batchrun import BatchedVanillaBO
    torch.set_num_threads(4)
    insts = list(range(B))
    seeds = [1000 * fid + 10 * dim + i for i in insts]
    r = BatchedVanillaBO([BBOBProblem(fid, i, dim) for i in insts], seeds, budget, n_doe, acq_kernel="latency")
    r.run()
    for b, i in enumerate(insts):
        X, f = _single_vanilla(fid, i, dim, budget, n_doe, seeds[b])
        assert np.array_equal(np.vstack(r.x_evals[b]), X), (fid, b)
        assert np.array_equal(np.array(r.f_evals[b]), f), (fid, b)
    runs = {}
    for mode in ("device", "device-twin"):
        runs[mode] = BatchedVanillaBO([BBOBProblem(fid, i, dim) for i in insts], seeds, budget, n_doe, acq_kernel=mode)
        runs[mode].run()
    for b in range(B):
        assert np.array_equal(np.vstack(runs["device"].x_evals[b]), np.vstack(runs["device-twin"].x_evals[b])), (fid, b)
        assert len(runs["device"].f_evals[b]) == budget


def test_experiment_runner_batches_vanilla_too(native, tmp_path):
    """main.py's default experiment runs both algorithms: with `batched` the Vanilla_BO runs advance in lock-step as well and
    the IOHprofiler files are the ones the serial runs write."""
    import os
    from Algorithms import ExperimentRunner
    outs = []
    for tag, batched in (("serial", 0), ("batched", 3)):
        root = tmp_path / tag
        er = ExperimentRunner(algorithms=["vanilla"], dimensions=[5], problem_ids=[15, 20], num_runs=3, budget_factor=5,
                              doe_factor=2.0, root_dir=str(root), experiment_name="experiment", progress=False, batched=batched,
                              batch_acq_kernel="latency")
        er.run_experiment()
        assert len(er.results) == 6
        outs.append((root, sorted((r["problem_id"], r["instance"], r["best"]) for r in er.results)))
    assert outs[0][1] == outs[1][1]
    for fid, name in ((15, "RastriginRotated"), (20, "Schwefel")):
        rel = os.path.join("vanilla-experiment", f"data_f{fid}_{name}", f"IOHprofiler_f{fid}_DIM5.dat")
        a, b = open(os.path.join(outs[0][0], rel)).read(), open(os.path.join(outs[1][0], rel)).read()
        assert a == b
