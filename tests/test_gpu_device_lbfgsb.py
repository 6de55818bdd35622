"""Device-resident L-BFGS-B of the batched path (first option; csrc/kernels_lbfgsb.hip): every restart
group's whole optimisation (botorch gen_candidates_scipy -> scipy L-BFGS-B, PCA_BO.py:607-614) inside ONE kernel launch.

  * the device steps must be the host's (csrc/lbfgsb.cpp, pinned against scipy on the CPU): whole runs with the stepping on the
    device equal, bit for bit, the same runs with the host stepping the same evaluation kernel launch by launch ("device-twin");
  * the evaluation (a third summation order) against the oracle: value + gradient at teacher-forced states;
  * runs in device mode replayed by the oracle with the thresholds of the host-paced batches.
"""
import numpy as np
import pytest
import torch

import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
from test_gpu_configs import _seed, _teacher_force
from test_gpu_late_phase import _rel
from test_gpu_parity import _check_replay, _replay_with_oracle

pytestmark = pytest.mark.gpu


def _run(fid, insts, dim, budget, n_doe, acq_kernel, **kw):
    from pcabo.batchrun import BatchedPCABO
    r = BatchedPCABO([BBOBProblem(fid, i, dim) for i in insts], [_seed(fid, dim, i) for i in insts], budget, n_doe,
                     acq_kernel=acq_kernel, **kw)
    r.run()
    return r


@pytest.mark.parametrize("fid,dim,budget,n_doe,B", [(15, 10, 90, 30, 5), (17, 20, 250, 60, 4), (15, 40, 450, 120, 3)])
def test_device_stepping_equals_host_stepping_bit_for_bit(native, fid, dim, budget, n_doe, B):
    """Whole runs (full budgets at d = 20 / 40: n up to 249 / 449, every row-split of the evaluation from 8 parts at n <= 128
    to 2 at n > 341): L-BFGS-B inside the kernel against lbfgsb.cpp on the host over the same evaluation kernel.  Every
    evaluated point, every objective value and the optimiser's counters of every iteration must be identical - L-BFGS-B
    amplifies a one-ulp difference by ~10 per five evaluations (tests/test_lbfgsb_divergence.py), so nothing short of the same
    operations in the same order passes."""
    torch.set_num_threads(4)
    insts = list(range(B))
    keep = lambda b, n: True
    dev = _run(fid, insts, dim, budget, n_doe, "device", record_trace=True, trace_filter=keep)
    twin = _run(fid, insts, dim, budget, n_doe, "device-twin", record_trace=True, trace_filter=keep)
    assert dev.failed == twin.failed == [None] * B
    rounds = 0
    for b in range(B):
        assert np.array_equal(np.vstack(dev.x_evals[b]), np.vstack(twin.x_evals[b])), (dim, b)
        assert np.array_equal(np.array(dev.f_evals[b]), np.array(twin.f_evals[b])), (dim, b)
    assert len(dev.trace) == len(twin.trace) == B * (budget - n_doe)
    for td, tt in zip(dev.trace, twin.trace):
        assert (td["b"], td["n"]) == (tt["b"], tt["n"])
        assert np.array_equal(td["cands"], tt["cands"]) and np.array_equal(td["vals"], tt["vals"]), (td["b"], td["n"])
        assert np.array_equal(td["info"], tt["info"]), (td["b"], td["n"], td["info"], tt["info"])
        rounds += int(np.asarray(td["info"])[:, 1].sum())
    print("[device = twin, f%d d=%d] %d runs x %d iterations, %d L-BFGS-B evaluations compared" % (fid, dim, B, budget - n_doe, rounds))


@pytest.mark.parametrize("dim,budget,n_doe,states", [(20, 250, 60, (61, 130, 249)), (40, 450, 120, (121, 260, 350, 420, 449))])
def test_device_evaluation_against_oracle(native, dim, budget, n_doe, states):
    """Value and gradient of the device optimiser's evaluation (pass 1 over the transposed root inverse, pass 2 over the root
    inverse, work plans of 1 .. 8 slabs) at states of a device-mode run, against the oracle's torch-autograd log-EI; the
    values the run itself reported for its end points against the oracle's surface."""
    torch.set_num_threads(4)
    fid, inst = 15, 1
    r = _run(fid, [inst, inst + 1], dim, budget, n_doe, "device", record_trace=True, trace_filter=lambda b, n: b == 0 and n in states)
    X, f = np.vstack(r.x_evals[0]), np.array(r.f_evals[0], dtype=float)
    assert len(r.trace) == len(states)
    worst = {"val": 0.0, "grad": 0.0, "run_vals": 0.0}
    for tr in r.trace:
        rec = _teacher_force(X, f, tr, BBOBProblem(fid, inst, dim), dim)
        n, k = rec.n, rec.k
        bt = native.Batch(1, max_n=budget, max_d=dim, max_q=512, device_lbfgsb=1)
        bt.wpca_gp_condition_begin(rec.X[None], rec.ranks[None], rec.noise[None], np.asarray(rec.f, dtype=float)[None])
        res = bt.wpca_results()
        assert res[0]["k"] == k == tr["k"]
        vraw, st = bt.gp_wait_eval([rec.trace.raw_X], [rec.best_f])
        assert not st.any()
        gp = O.ExactGP(rec.wpca.Z, rec.f, rec.norm_bounds)
        acq = O.Acquisition(gp, rec.best_f, False)
        Xs = np.vstack([rec.trace.ics, rec.trace.cands, rec.trace.raw_X[:12]])          # 32 queries
        ov, og = acq.value_and_grad(Xs)
        v, g = bt.device_acq_eval([Xs], [rec.best_f])
        worst["val"] = max(worst["val"], float((np.abs(v[0] - ov) / np.maximum(1.0, np.abs(ov))).max()))
        worst["grad"] = max(worst["grad"], _rel(g[0], og))
        # the per-query kernels (k_acq_fast: DPP tree sums) on the SAME factor R at the same points: what part of the distance to
        # the oracle is the evaluation's summation order, what part the factorisation both share
        bt.ctx[0].k, bt.ctx[0].n = k, n                     # (the batch's borrowed context: its shape is the batch's)
        vq, gq = bt.ctx[0].acq_eval(Xs, rec.best_f)
        worst["val_per_query"] = max(worst.get("val_per_query", 0.0), float((np.abs(vq - ov) / np.maximum(1.0, np.abs(ov))).max()))
        worst["grad_per_query"] = max(worst.get("grad_per_query", 0.0), _rel(gq, og))
        worst["val_between"] = max(worst.get("val_between", 0.0), float((np.abs(vq - v[0]) / np.maximum(1.0, np.abs(vq))).max()))
        worst["grad_between"] = max(worst.get("grad_between", 0.0), _rel(g[0], gq))
        v7, g7 = bt.device_acq_eval([Xs[:7]], [rec.best_f])           # a point's numbers do not depend on its group
        assert np.array_equal(v7[0], v[0][:7]) and np.array_equal(g7[0], g[0][:7])
        vo = rec.acq(torch.from_numpy(np.ascontiguousarray(tr["cands"]))).detach().numpy()
        worst["run_vals"] = max(worst["run_vals"], float((np.abs(vo - tr["vals"]) / np.maximum(1.0, np.abs(tr["vals"]))).max()))
        del bt
    print("[device evaluation vs oracle, d=%d] value %.2e gradient %.2e, the run's own end-point values %.2e; the per-query kernels at the "
          "same points: value %.2e gradient %.2e; the two kernels against each other: value %.2e gradient %.2e" % (
        dim, worst["val"], worst["grad"], worst["run_vals"], worst["val_per_query"], worst["grad_per_query"], worst["val_between"],
        worst["grad_between"]))
    # measured (round 4, profiles/r04/device_eval_accuracy.txt): the two kernels agree with EACH OTHER to ~1e-13 on the same factor -
    # their distance to the oracle at these points (restart end points: the posterior variance is tiny there, 1 - |v|^2 cancels) is
    # the distance between the device's factorisation and torch's, shared by every kernel.  Thresholds: measured x 10.
    assert worst["val"] < 5e-9 and worst["grad"] < 3e-10 and worst["run_vals"] < 5e-9, worst
    assert worst["val_between"] < 1e-10 and worst["grad_between"] < 1e-10, worst


def test_device_mode_batch_replayed_by_oracle(native):
    """A 12-run device-mode batch of a configs[2] cell (f16, d = 20, full budget); two runs replayed by the oracle from
    their own states: same k, same raw-sample picks, restart end points, counts, chosen candidate and objective value
    (thresholds of the host-paced batches, tests/test_gpu_configs.py)."""
    from pcabo.batchrun import BatchedPCABO
    torch.set_num_threads(4)
    fid, dim, B, every = 16, 20, 12, 24
    budget, n_doe = 10 * dim + 50, 3 * dim
    replayed = (0, 7)
    r = BatchedPCABO([BBOBProblem(fid, i, dim) for i in range(B)], [_seed(fid, dim, i) for i in range(B)], budget, n_doe,
                     record_trace=True, trace_filter=lambda b, n: b in replayed and (n - n_doe) % every == 2, acq_kernel="device")
    r.run()
    assert all(x is None for x in r.failed), r.failed
    for b in range(B):
        assert len(r.f_evals[b]) == budget and r.current_best[b] == min(r.f_evals[b])

    class _View:
        pass

    for b in replayed:
        v = _View()
        v.x_evals, v.f_evals, v.maximization = r.x_evals[b], r.f_evals[b], False
        v.trace = [t for t in r.trace if t["b"] == b]
        st = _replay_with_oracle(v, lambda: BBOBProblem(fid, b, dim), dim)
        _check_replay(st, min_iters=len(v.trace) - 1, late=True)


def test_device_mode_falls_back_where_it_does_not_apply(native):
    """k > 40 is beyond the device optimiser's LDS layout: the call takes the host-paced path and returns what the batch
    returns without the option."""
    rng = np.random.default_rng(5)
    B, n, d, q = 2, 180, 60, 64
    X = rng.uniform(-5, 5, (B, n, d))
    y = rng.normal(size=(B, n)) * 50 + 300
    ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
    outs = []
    for dev in (0, 1):
        bt = native.Batch(B, max_n=200, max_d=d, max_q=q, device_lbfgsb=dev)
        bt.wpca_gp_condition_begin(X, ranks, None, y, n_components=48)
        res = bt.wpca_results()
        assert min(r["k"] for r in res) > 40
        boxes = bt.acq_bounds()
        raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * np.random.default_rng(b).uniform(size=(q, res[b]["k"])) for b in range(B)]
        best = [float(y[b].min()) for b in range(B)]
        vals, status = bt.gp_wait_eval(raw, best)
        o, st = bt.optimize_acqf([raw[b][:10] for b in range(B)], boxes, best)
        assert not st.any()
        outs.append(o)
        del bt
    for b in range(B):
        assert np.array_equal(outs[0][b][0], outs[1][b][0]) and np.array_equal(outs[0][b][2], outs[1][b][2])


def test_interleaved_batches_equal_batches_run_alone(native):
    """pcabo.batchrun.run_interleaved: three device-mode batches advanced from ONE host thread (begin / end halves of the
    scoring and of the optimisation, pcabo_batch_busy) - every run must take, bit for bit, the path it takes in a batch that
    runs alone with the blocking calls."""
    from pcabo.batchrun import BatchedPCABO, run_interleaved
    torch.set_num_threads(4)
    fid, dim, budget, n_doe = 15, 10, 80, 30
    groups = [[0, 1, 2], [3, 4], [5, 6, 7, 8]]

    def make(insts):
        return BatchedPCABO([BBOBProblem(fid, i, dim) for i in insts], [_seed(fid, dim, i) for i in insts], budget, n_doe,
                            acq_kernel="device")
    together = [make(g) for g in groups]
    run_interleaved(together)
    for g, rt in zip(groups, together):
        alone = make(g)
        alone.run()
        for b in range(len(g)):
            assert np.array_equal(np.vstack(rt.x_evals[b]), np.vstack(alone.x_evals[b])), (g, b)
            assert np.array_equal(np.array(rt.f_evals[b]), np.array(alone.f_evals[b])), (g, b)
            assert len(rt.f_evals[b]) == budget


def test_device_mode_blown_up_run_and_parked_run(native):
    """The reference's unclipped candidates blow the search box of f21 / instance 25 at d = 40 up to |x| ~ 1e55
    (tests/test_gpu_configs.py::test_blown_up_run_follows_the_oracle_to_the_end): the device-resident optimiser must carry such a
    state like the host does - same path as the host-stepped twin, bit for bit, nothing parked; and a run parked by the caller
    must not disturb the other runs of a device-mode batch."""
    from pcabo.batchrun import BatchedPCABO
    torch.set_num_threads(4)
    fid, inst, dim, budget, n_doe = 21, 25, 40, 450, 120
    runs = {}
    for mode in ("device", "device-twin"):
        r = BatchedPCABO([BBOBProblem(fid, inst, dim), BBOBProblem(15, 0, dim)], [_seed(fid, dim, inst), _seed(15, dim, 0)],
                         budget, n_doe, acq_kernel=mode)
        r.run()
        assert r.failed == [None, None]
        runs[mode] = r
    X = np.vstack(runs["device"].x_evals[0])
    assert np.abs(X).max() > 1e40 and np.isfinite(X).all()
    for b in range(2):
        assert np.array_equal(np.vstack(runs["device"].x_evals[b]), np.vstack(runs["device-twin"].x_evals[b])), b
        assert np.array_equal(np.array(runs["device"].f_evals[b]), np.array(runs["device-twin"].f_evals[b])), b
    # a parked run: the others take the path they take without it
    insts = [0, 1, 2]
    seeds = [15100 + i for i in insts]
    r = BatchedPCABO([BBOBProblem(15, i, 10) for i in insts], seeds, 60, 30, acq_kernel="device")
    r.start()
    for it in range(30):
        if it == 7:
            r._park(1, r.n, "parked by the test")
        r.iteration()
    r.finish()
    alone = BatchedPCABO([BBOBProblem(15, i, 10) for i in (0, 2)], [seeds[0], seeds[2]], 60, 30, acq_kernel="device")
    alone.run()
    for b, a in ((0, 0), (2, 1)):
        assert np.array_equal(np.vstack(r.x_evals[b]), np.vstack(alone.x_evals[a])), b


def test_experiment_runner_in_device_mode(native, tmp_path):
    """ExperimentRunner(batch_acq_kernel="device"): the reference's runner surface over device-mode batches interleaved on one
    host thread; the results and the IOHprofiler files do not depend on how the runs are grouped into batches."""
    import os
    from Algorithms import ExperimentRunner
    outs = []
    for tag, batched, sbs in (("a", 3, 2), ("b", 6, 1)):
        root = tmp_path / tag
        er = ExperimentRunner(algorithms=["pca"], dimensions=[5], problem_ids=[15, 20], num_runs=3, budget_factor=5, doe_factor=2.0,
                              root_dir=str(root), experiment_name="experiment", progress=False, batched=batched, side_by_side=sbs,
                              batch_acq_kernel="device")
        er.run_experiment()
        assert len(er.results) == 6 and not er.failed_runs
        outs.append((root, sorted((r["problem_id"], r["instance"], r["best"]) for r in er.results)))
    assert outs[0][1] == outs[1][1]
    for fid, name in ((15, "RastriginRotated"), (20, "Schwefel")):
        rel = os.path.join("pca-experiment", f"data_f{fid}_{name}", f"IOHprofiler_f{fid}_DIM5.dat")
        assert open(os.path.join(outs[0][0], rel)).read() == open(os.path.join(outs[1][0], rel)).read()


def test_device_stepping_with_probability_of_improvement(native):
    """The other acquisition of the reference's surface (PCA_BO.py:653-677: "probability_of_improvement") through the device-resident
    optimiser: same bits as the host-stepped twin (PI's gradient has no sigma term of its own: another branch of the scalar chain)."""
    torch.set_num_threads(4)
    fid, dim, budget, n_doe, B = 15, 10, 70, 30, 3
    dev = _run(fid, list(range(B)), dim, budget, n_doe, "device", acquisition_function="probability_of_improvement")
    twin = _run(fid, list(range(B)), dim, budget, n_doe, "device-twin", acquisition_function="probability_of_improvement")
    for b in range(B):
        assert np.array_equal(np.vstack(dev.x_evals[b]), np.vstack(twin.x_evals[b])), b
        assert np.array_equal(np.array(dev.f_evals[b]), np.array(twin.f_evals[b])), b
        assert len(dev.f_evals[b]) == budget


def test_auto_mode_is_one_trajectory_per_seed_on_any_number_of_ranks(native, tmp_path, monkeypatch):
    """ExperimentRunner(batch_acq_kernel="auto") end to end on a d = 20 cell with 30 runs (ADVICE):
    `auto` resolves to the device-resident optimiser from the EXPERIMENT's description, so
      * the files equal those of an explicit "device" experiment byte for byte,
      * the same experiment cut into the shares of two ranks (each share 15 runs - below the 30 that `auto` looks for, had it
        looked at the share) writes the same rows per run, and
      * the mode is written into the experiment attributes of every file."""
    import os
    from Algorithms import ExperimentRunner
    from pcabo import iohlog
    common = dict(algorithms=["pca"], dimensions=[20], problem_ids=[15], num_runs=30, budget_factor=4, doe_factor=3.0,
                  experiment_name="experiment", progress=False, batched=30, side_by_side=2)
    rel = os.path.join("data_f15_RastriginRotated", "IOHprofiler_f15_DIM20.dat")

    def blocks(folder):
        return iohlog.read_dat(os.path.join(folder, rel))

    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(var, raising=False)
    er = ExperimentRunner(root_dir=str(tmp_path / "auto"), batch_acq_kernel="auto", **common)
    assert er.arithmetic_modes == {20: "device"}
    er.run_experiment()
    assert len(er.results) == 30 and not er.failed_runs
    er_dev = ExperimentRunner(root_dir=str(tmp_path / "device"), batch_acq_kernel="device", **common)
    er_dev.run_experiment()
    a = open(os.path.join(str(tmp_path / "auto"), "pca-experiment", rel)).read()
    assert a == open(os.path.join(str(tmp_path / "device"), "pca-experiment", rel)).read()
    js = open(os.path.join(str(tmp_path / "auto"), "pca-experiment", "IOHprofiler_f15_RastriginRotated.json")).read()
    assert '"arithmetic_mode": "d20=device"' in js
    # the phase timers are shares of the run's time (ADVICE: they used to sum to a multiple of it in interleaved mode)
    for r in er.results:
        assert r["pca"] + r["optimize_acqf"] + r["SingleTaskGP"] <= r["time"] * (1 + 1e-9)
    whole = {tuple(b[0, 3:]): b for b in blocks(os.path.join(str(tmp_path / "auto"), "pca-experiment"))}     # keyed by the first DoE point
    assert len(whole) == 30
    seen = 0
    for rank in range(2):
        monkeypatch.setenv("RANK", str(rank)); monkeypatch.setenv("LOCAL_RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2")
        part = ExperimentRunner(root_dir=str(tmp_path / "ranks"), batch_acq_kernel="auto", **common)
        assert part.arithmetic_modes == {20: "device"} and len(part._my_runs()) == 15
        part.run_experiment()
        for b in blocks(os.path.join(str(tmp_path / "ranks"), f"pca-experiment-rank{rank}")):
            assert np.array_equal(b, whole[tuple(b[0, 3:])])
            seen += 1
    assert seen == 30


def test_device_option_switched_on_after_a_conditioning(native):
    """PCABO_OPT_DEVICE_LBFGSB switched on AFTER a conditioning - the transposed root inverse does not exist
    yet (the Gram buffer holds K); the next evaluation must build it instead of reading whatever is there.  Same bits as a batch
    that had the option from the start; and the evaluation refuses to run between the halves of another call."""
    from pcabo import _native as N
    rng = np.random.default_rng(11)
    B, n, d, q = 2, 150, 12, 64
    X = rng.uniform(-5, 5, (B, n, d))
    y = (X ** 2).sum(axis=2) + rng.normal(size=(B, n))
    ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
    noise = rng.normal(0, 1e-8, (B, n, d))
    out = []
    for late in (False, True):
        bt = N.Batch(B, max_n=n, max_d=d, max_q=q, device_lbfgsb=0 if late else 1)
        bt.wpca_gp_condition_begin(X, ranks, noise, y)
        res = bt.wpca_results()
        boxes = bt.acq_bounds()
        raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * np.random.default_rng(5 + b).uniform(size=(q, res[b]["k"])) for b in range(B)]
        best = [float(y[b].min()) for b in range(B)]
        bt.gp_wait_eval(raw, best)
        if late:
            bt._chk(N.LIB.pcabo_batch_set_option(bt._h, N.OPT_DEVICE_LBFGSB, 1))
            bt.device_lbfgsb = 1
        out.append(bt.device_acq_eval([r[:10] for r in raw], best))
    for b in range(B):
        assert np.array_equal(out[0][0][b], out[1][0][b]) and np.array_equal(out[0][1][b], out[1][1][b])
        assert np.isfinite(out[0][0][b]).all()
