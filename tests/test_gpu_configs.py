"""BASELINE.json configs[2], configs[3] (its one-GPU point) and configs[4] under `-m gpu`, each checked against the CPU oracle.

  configs[2]  f15/f16/f17 x d in {10, 20, 40} x 30 runs on one GPU       -> test_configs2_30_run_batches_replayed_by_oracle
  configs[3]  30 runs x f15..f24 x d in {20, 40}, runs sharded over GPUs -> test_configs3_functions_f16_to_f24_states_against_oracle,
              (reference: ExperimentRunner.py:90,137-183)                   test_blown_up_run_follows_the_oracle_to_the_end
  configs[4]  d = 100, doe_factor 3, 256 EI multi-starts                 -> test_configs4_256_restarts_d100_against_oracle

The device runs are the product's own lock-step batches (`pcabo.batchrun.BatchedPCABO`, k_acq_group for the L-BFGS-B rounds);
at sampled (run, n) the oracle is teacher-forced from the run's state (same X, f, numpy / torch generator states) and the
device - through the C ABI - must reproduce what the oracle computes from those inputs (helpers of test_gpu_late_phase.py).
"""
import numpy as np
import pytest
import torch

import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
from test_gpu_late_phase import _check_state, _rel
from test_gpu_parity import _check_replay, _replay_with_oracle

pytestmark = pytest.mark.gpu

FIDS = tuple(range(16, 25))


def _seed(fid, dim, inst):
    return 1000 * fid + 10 * dim + inst          # ExperimentRunner.py:146


def _teacher_force(X, f, tr, problem, dim):
    """One oracle iteration from a device run's state: design matrix, objective values and the run's generator states."""
    n = tr["n"]
    orc = O.OraclePCABO(budget=n + 1, n_DoE=n, random_seed=0, maximization=False, record=True)
    orc.x_evals = [row.copy() for row in X[:n]]
    orc.f_evals = [float(v) for v in f[:n]]
    orc._assign_new_best()
    assert orc.current_best == tr["best_f"]
    np.random.set_state(tr["numpy_state"])
    torch.set_rng_state(tr["torch_state"])
    return orc.step(problem, np.full(dim, -5.0), np.full(dim, 5.0))


def _state_n(dim, fid):
    """One state per (function, dimension), spread over the BO phase: d=20: n = 70..214, d=40: n = 130..370."""
    return (70 + 18 * (fid - 16)) if dim == 20 else (130 + 30 * (fid - 16))


@pytest.mark.parametrize("dim", [20, 40])
def test_configs3_functions_f16_to_f24_states_against_oracle(native, dim):
    """f16..f24 go through the BO loop on the device (one lock-step batch of the nine functions, instance 0, seeds as the
    reference's runner assigns them); one state per function is then checked at kernel level against the oracle: wPCA, GP
    state, value + gradient (per-query kernels and k_acq_group), the 512 raw scores, finite differences, one
    optimize_acqf in both launch modes, the inverse map, the same raw-sample picks."""
    from pcabo.batchrun import BatchedPCABO
    torch.set_num_threads(4)
    budget, n_doe = 10 * dim + 50, 3 * dim
    want = {b: _state_n(dim, fid) for b, fid in enumerate(FIDS)}
    r = BatchedPCABO([BBOBProblem(fid, 0, dim) for fid in FIDS], [_seed(fid, dim, 0) for fid in FIDS], budget, n_doe,
                     record_trace=True, trace_filter=lambda b, n: want[b] == n)
    r.start()
    try:
        while r.n <= max(want.values()):
            r.iteration()
    finally:
        r.finish()
    assert all(f is None for f in r.failed), r.failed
    assert sorted(t["b"] for t in r.trace) == list(range(len(FIDS)))
    ctx = native.Context(max_n=budget, max_d=dim, max_q=512)
    stats = []
    for tr in sorted(r.trace, key=lambda t: t["b"]):
        b, fid = tr["b"], FIDS[tr["b"]]
        X, f = np.vstack(r.x_evals[b]), np.array(r.f_evals[b], dtype=float)
        rec = _teacher_force(X, f, tr, BBOBProblem(fid, 0, dim), dim)
        s = _check_state(native, ctx, rec, tr, stats)
        s["fid"] = fid
        # the batched run's own iteration at this state (k_acq_group rounds inside the batch): its end points judged by
        # the oracle's surface, and its chosen candidate against what the run then evaluated
        vo = rec.acq(torch.from_numpy(np.ascontiguousarray(tr["cands"]))).detach().numpy()
        assert float((np.abs(vo - tr["vals"]) / np.maximum(1.0, np.abs(tr["vals"]))).max()) < 1e-8, (fid, dim)
        x_dev = ctx.inverse_map(tr["cands"][tr["chosen"]])           # (ctx holds the oracle-input wPCA of this state)
        assert _rel(x_dev, X[tr["n"]]) < 1e-8, (fid, dim)
    ctx.close()
    assert {s["fid"] for s in stats} == set(FIDS)
    counts = [c for s in stats for c in s["counts_equal"]]
    cands = [c for s in stats for c in s.get("cand", [])]
    print("[configs3 d=%d] states %d: counts equal %.3f; end points median %.2e <1e-5 %.3f; worst kernel-level errors: Z %.1e K %.1e L %.1e "
          "alpha %.1e value %.1e grad %.1e raw %.1e" % (dim, len(stats), np.mean(counts), np.median(cands), np.mean(np.array(cands) < 1e-5),
                                                       *[max(s[key] for s in stats) for key in ("Z", "K", "L", "alpha", "val", "grad", "raw")]))
    # optimiser statistics as in the late-phase test of the headline run (kernel-level agreement is asserted per state)
    # (measured round 3: counts equal 0.78 (d=20) / 0.83 (d=40); end points median 0 / 4e-13, within 1e-5 0.80 / 0.83)
    assert np.mean(counts) >= 0.70, counts
    assert np.median(cands) < 1e-9 and np.mean(np.array(cands) < 1e-5) >= 0.72, np.sort(cands)[-10:]


@pytest.mark.parametrize("fid,dim,every", [(16, 20, 24), (17, 40, 44)])
def test_configs2_30_run_batches_replayed_by_oracle(native, fid, dim, every):
    """One 30-run lock-step batch of a configs[2] cell (30 instances, full budget); three of its runs are replayed by the
    oracle from their own states (every `every`-th iteration): same k, same raw-sample picks, restart end points, counts,
    chosen candidate and objective value (thresholds of the free-running replays of test_gpu_parity.py)."""
    from pcabo.batchrun import BatchedPCABO
    torch.set_num_threads(4)
    budget, n_doe, B = 10 * dim + 50, 3 * dim, 30
    replayed = (0, 13, 29)
    r = BatchedPCABO([BBOBProblem(fid, i, dim) for i in range(B)], [_seed(fid, dim, i) for i in range(B)], budget, n_doe,
                     record_trace=True, trace_filter=lambda b, n: b in replayed and (n - n_doe) % every == 2)
    r.run()
    assert all(f is None for f in r.failed), r.failed
    for b in range(B):
        assert len(r.f_evals[b]) == budget and r.current_best[b] == min(r.f_evals[b])

    class _View:
        pass

    total = {"iters": 0}
    for b in replayed:
        v = _View()
        v.x_evals, v.f_evals, v.maximization = r.x_evals[b], r.f_evals[b], False
        v.trace = [t for t in r.trace if t["b"] == b]
        assert len(v.trace) >= (budget - n_doe) // every
        st = _replay_with_oracle(v, lambda: BBOBProblem(fid, b, dim), dim)
        _check_replay(st, min_iters=len(v.trace) - 1, late=True)
        total["iters"] += st["iters"]
    assert total["iters"] >= 3 * ((budget - n_doe) // every) - 3


def test_blown_up_run_follows_the_oracle_to_the_end(native):
    """The reference keeps out-of-box candidates (penalised, not clipped: PCA_BO.py:253,260-263); each widens the next search
    box by half (PCA_BO.py:558-573), so on some functions of configs[3] (f19, f21-f24 at d = 40) coordinates grow
    geometrically - 1e55 and beyond by the end of the budget, k collapsed to 1.  sklearn / torch carry such a state (LAPACK
    scales the covariance), so the reference's run goes on to its budget, and so must the device's.  Round 2's device run
    of f21 / instance 25 stopped at n = 412 with PCABO_ERR_NAN: its Jacobi sweep squared covariance entries of ~1e160 and
    left the double range (the oracle, teacher-forced from that very state, raised nothing - this test, round 3); the
    sweep now scales the matrix by a power of four first.  Here: the device run reaches its budget, and the oracle
    replays the iterations around the first |x| > 1e30 state and the last six from the device's own states - same k,
    same raw-sample picks, same end points, same chosen candidate; a batch holding the run parks nothing."""
    from Algorithms import PCA_BO
    from pcabo.batchrun import BatchedPCABO
    torch.set_num_threads(4)
    fid, inst, dim = 21, 25, 40
    budget, n_doe = 450, 120
    opt = PCA_BO(budget=budget, n_DoE=n_doe, random_seed=_seed(fid, dim, inst), maximization=False, record_trace=True,
                 acq_kernel="group")
    opt(BBOBProblem(fid, inst, dim))
    assert len(opt.f_evals) == budget
    X, f = np.vstack(opt.x_evals), np.array(opt.f_evals, dtype=float)
    assert np.abs(X).max() > 1e40 and np.isfinite(X).all()                 # the blow-up the reference's dynamics produce
    first_big = next(t["n"] for t in opt.trace if np.abs(X[:t["n"]]).max() > 1e30)
    checked = 0
    for tr in opt.trace:
        n = tr["n"]
        if not (first_big <= n < first_big + 2 or n >= budget - 6):
            continue
        rec = _teacher_force(X, f, tr, BBOBProblem(fid, inst, dim), dim)      # (raises if the oracle meets a NaN gradient)
        assert rec.k == tr["k"], n
        if rec.trace.retried or tr.get("retried", False):
            assert rec.trace.retried == bool(tr.get("retried", False)), n
            continue
        assert sorted(rec.trace.ic_idx.tolist()) == sorted(tr["ic_idx"].tolist()), n
        scale = max(1.0, np.abs(rec.trace.cands).max())
        assert np.abs(rec.trace.cands - tr["cands"]).max() < 1e-8 * scale, n
        assert np.abs(rec.cand_x - X[n]).max() < 1e-8 * max(1.0, np.abs(rec.cand_x).max()), n
        vo = rec.acq(torch.from_numpy(np.ascontiguousarray(tr["cands"]))).detach().numpy()
        assert float((np.abs(vo - tr["vals"]) / np.maximum(1.0, np.abs(tr["vals"]))).max()) < 1e-8, n
        checked += 1
    assert checked >= 6
    # the lock-step batch: the same run beside a tame one - nothing parked, and the run takes the path it takes alone
    r = BatchedPCABO([BBOBProblem(fid, inst, dim), BBOBProblem(15, 0, dim)], [_seed(fid, dim, inst), _seed(15, dim, 0)],
                     budget, n_doe)
    r.run()
    assert r.failed == [None, None]
    assert np.array_equal(np.vstack(r.x_evals[0]), X) and np.array_equal(np.array(r.f_evals[0]), f)


def test_configs4_256_restarts_d100_against_oracle(native):
    """configs[4]: d = 100, n_DoE = 300 (k ~ 85), 256 EI multi-starts = 52 joint L-BFGS-B problems.  Device call against the
    oracle's gen_candidates_scipy on the same 256 initial conditions: per group counts (small slack: 5 k = 425 joint
    variables), end points, values; the chosen candidate; the device surface at all 256 device end points."""
    torch.set_num_threads(8)
    o = O.OraclePCABO(budget=1050, n_DoE=300, random_seed=_seed(15, 100, 0), record=True)
    o(BBOBProblem(15, 0, 100), 100, np.array([-5.0, 5.0]), max_iters=1)
    rec = o.records[0]
    c = native.Context(max_n=1050, max_d=100, max_q=512)
    res = c.wpca(rec.X, ranks=rec.ranks, noise=rec.noise)
    assert res["k"] == rec.k and rec.k > 64
    c.gp_condition(rec.f)
    gp = O.ExactGP(rec.wpca.Z, rec.f, rec.norm_bounds)
    acq = O.Acquisition(gp, rec.best_f, False)
    ics = rec.trace.raw_X[np.argsort(-rec.trace.raw_vals, kind="stable")[:256]]
    cand, vals, info, failed = c.optimize_acqf(ics, rec.acq_bounds, rec.best_f)
    assert info.shape == (52, 4) and not failed
    scale = max(1.0, np.abs(ics).max())
    ocand, ovals, same_counts, near_counts = [], [], 0, 0
    for g in range(52):
        sl = slice(5 * g, min(256, 5 * g + 5))
        oc, ov, ofail, t = O.gen_candidates_scipy(ics[sl], acq, rec.acq_bounds)
        assert not ofail
        ocand.append(oc), ovals.append(ov)
        same_counts += (t.nit, t.nfev) == (int(info[g, 0]), int(info[g, 1]))
        near_counts += abs(t.nit - int(info[g, 0])) <= 2 and abs(t.nfev - int(info[g, 1])) <= 3
    ocand, ovals = np.vstack(ocand), np.concatenate(ovals)
    dc = np.abs(cand - ocand).max(axis=1) / scale
    dv = np.abs(vals - ovals) / np.maximum(1.0, np.abs(ovals))
    # measured on MI355X (round 3): counts identical in 52 of 52 groups, end points max 1.2e-5 (99.6 % within 1e-5), values all
    # within 1e-6 - thresholds = measured with a margin
    print("[configs4 256 restarts] counts identical %d / 52, within (2, 3) %d; end points median %.2e <1e-5 %.3f max %.2e; values median "
          "%.2e <1e-6 %.3f" % (same_counts, near_counts, np.median(dc), np.mean(dc < 1e-5), dc.max(), np.median(dv), np.mean(dv < 1e-6)))
    assert same_counts >= 47 and near_counts >= 49, (same_counts, near_counts)
    assert np.median(dc) < 1e-9 and np.mean(dc < 1e-5) >= 0.9 and dc.max() < 1e-3, np.sort(dc)[-8:]
    assert np.median(dv) < 1e-12 and np.mean(dv < 1e-6) >= 0.9, np.sort(dv)[-8:]
    vo = acq(torch.from_numpy(np.ascontiguousarray(cand))).detach().numpy()
    assert float((np.abs(vo - vals) / np.maximum(1.0, np.abs(vals))).max()) < 1e-8
    bo, bd = int(np.argmax(ovals)), int(np.argmax(vals))
    if bo != bd:                                           # another restart may win only by a numerical tie
        assert abs(ovals[bo] - ovals[bd]) < 1e-7 * max(1.0, abs(ovals[bo])), (bo, bd, ovals[bo], ovals[bd])
    else:
        x = c.inverse_map(cand[bd])
        xo = O.inverse_map(ocand[bo], rec.wpca)
        assert np.abs(x - xo).max() < 1e-5 * max(1.0, np.abs(xo).max())
    c.close()
