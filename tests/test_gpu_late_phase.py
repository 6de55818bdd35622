"""GPU parity in the LATE phase of the headline run (BASELINE.json configs[1]: f15, d=40, n = 120 -> 449) and on d=20
states (configs[2]): every `k_acq_fast<SLAB,NB>` instantiation the benchmark executes is compared with the CPU oracle.

A free-running device run supplies the states (X, f and the numpy / torch RNG states before each iteration); at the
sampled n the oracle is teacher-forced from that state and the device - through the C ABI - must reproduce, from the
oracle's inputs: the weighted PCA, the GP state, acquisition value + gradient (q <= 32: in-launch finish), the scoring
of the 512 raw samples (q >= 64: 8 queries per group + combine pass), one multi-start L-BFGS-B call (iteration and
evaluation counts, end points) in BOTH launch modes (resident kernel fed through the mailbox; one launch per
evaluation), and a finite-difference check of the analytic gradient.

    n      NP   instantiation (plain <.., false> and resident <.., true>)
    120    128  <16,2>       256  256 <16,4>       384  384 <16,6>       448  448 <32,7>
    192    192  <16,3>       320  320 <16,5>       385  448 <32,7>       449  512 <32,8>
  d=20: 60 -> <16,1>, 128 -> <16,2>, 129 -> <16,3>, 200 -> <16,4>, 249 -> <16,4>

Tolerances are fp64 round-off amplified by the conditioning of the state (stated per assertion); north_star asks 1e-5.
"""
import numpy as np
import pytest
import torch

import pcabo_oracle as O
from pcabo.bbob import BBOBProblem

pytestmark = pytest.mark.gpu

HEADLINE_NS = (120, 192, 256, 320, 384, 385, 448, 449)
D20_NS = (60, 128, 129, 200, 249)


def _free_run(dim, budget, n_doe, seed, inst):
    from Algorithms import PCA_BO
    torch.set_num_threads(4)
    opt = PCA_BO(budget=budget, n_DoE=n_doe, random_seed=seed, maximization=False, record_trace=True)
    opt(BBOBProblem(15, inst, dim))
    assert len(opt.f_evals) == budget and len(opt.trace) == budget - n_doe
    return opt


@pytest.fixture(scope="module")
def headline_run(native):
    return _free_run(40, 450, 120, 15400, 0)


@pytest.fixture(scope="module")
def d20_run(native):
    return _free_run(20, 250, 60, 15200, 0)


def _oracle_step(opt, n, dim, inst):
    """Teacher-force the oracle from the device run's state at n evaluated points (same X, f, RNG states)."""
    X_all, f_all = np.vstack(opt.x_evals), np.array(opt.f_evals, dtype=float)
    tr = opt.trace[n - opt.n_DoE]
    assert tr["n"] == n
    orc = O.OraclePCABO(budget=n + 1, n_DoE=n, random_seed=0, maximization=False, record=True)
    orc.x_evals = [row.copy() for row in X_all[:n]]
    orc.f_evals = [float(v) for v in f_all[:n]]
    orc._assign_new_best()
    assert orc.current_best == tr["best_f"]
    np.random.set_state(tr["numpy_state"])
    torch.set_rng_state(tr["torch_state"])
    rec = orc.step(BBOBProblem(15, inst, dim), np.full(dim, -5.0), np.full(dim, 5.0))
    return rec, tr


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, float(np.abs(np.asarray(b)).max())))


def _check_state(native, ctx, rec, tr, stats):
    n, k = rec.n, rec.k
    s = {"n": n, "k": k}
    # ---- rows A-C: weighted PCA from the oracle's inputs -------------------------------------------------------
    res = ctx.wpca(rec.X, ranks=rec.ranks, noise=rec.noise)
    assert res["k"] == k == tr["k"], (n, res["k"], k, tr["k"])
    s["evr"] = _rel(res["evr"], rec.wpca.evr)
    s["Z"] = _rel(res["Z"], rec.wpca.Z)
    assert s["evr"] < 1e-10 and s["Z"] < 1e-8, s
    # ---- rows D-H ----------------------------------------------------------------------------------------------
    ctx.gp_condition(rec.f)
    st = ctx.gp_state()
    gp = O.ExactGP(rec.wpca.Z, rec.f, rec.norm_bounds)
    gp.condition()
    s["K"] = float(np.abs(ctx.gram() - gp.K.numpy()).max())
    s["L"] = float(np.abs(st["L"] - gp.L.numpy()).max())
    s["alpha"] = _rel(st["alpha"], gp.alpha.numpy())
    s["box"] = _rel(ctx.acq_bounds(), rec.acq_bounds)
    assert s["K"] < 1e-8 and s["L"] < 1e-7 and s["alpha"] < 1e-6 and s["box"] < 1e-9, s
    # ---- row I: value + gradient, small batch (in-launch finish) --------------------------------------------------
    acq = O.Acquisition(gp, rec.best_f, False)
    Xs = np.vstack([rec.trace.ics, rec.trace.cands, rec.trace.raw_X[:12]])          # 32 queries
    ov, og = acq.value_and_grad(Xs)
    v, g = ctx.acq_eval(Xs, rec.best_f, False)
    s["val"] = float((np.abs(v - ov) / np.maximum(1.0, np.abs(ov))).max())
    s["grad"] = _rel(g, og)
    assert s["val"] < 1e-8 and s["grad"] < 1e-6, s
    # the throughput kernel of the batched driver (k_acq_group: 5 queries per work-group, 64-row slabs) on the same points
    ctx.set_option(native.OPT_GROUP_ACQ, 1)
    vg, gg = ctx.acq_eval(Xs, rec.best_f, False)
    v7, g7 = ctx.acq_eval(Xs[:7], rec.best_f, False)            # a full group and one of two points
    ctx.set_option(native.OPT_GROUP_ACQ, 0)
    s["val_group"] = float((np.abs(vg - ov) / np.maximum(1.0, np.abs(ov))).max())
    s["grad_group"] = _rel(gg, og)
    assert s["val_group"] < 1e-8 and s["grad_group"] < 1e-6, s
    assert np.array_equal(v7, vg[:7]) and np.array_equal(g7, gg[:7])       # a point's numbers do not depend on its group
    # value-only path and single queries: the same arithmetic
    assert np.array_equal(ctx.acq_eval(Xs, rec.best_f, False, grad=False), v)
    v1, g1 = ctx.acq_eval(Xs[3:4], rec.best_f, False)
    assert np.array_equal(v1, v[3:4]) and np.array_equal(g1, g[3:4])
    # ---- rows K: the 512 raw samples (8 queries per group + combine pass) ----------------------------------------
    vr = ctx.acq_eval(rec.trace.raw_X, rec.best_f, False, grad=False)
    s["raw"] = float((np.abs(vr - rec.trace.raw_vals) / np.maximum(1.0, np.abs(rec.trace.raw_vals))).max())
    assert s["raw"] < 1e-8, s
    vb, gb = ctx.acq_eval(rec.trace.raw_X[:96], rec.best_f, False)                 # large batch WITH gradient
    ov96, og96 = acq.value_and_grad(rec.trace.raw_X[:96])
    s["grad_large"] = _rel(gb, og96)
    # (value-only batches of >= 64 points run as a GEMM on MFMA, batches with gradient through the slab kernels: same
    # numbers up to summation order)
    # (1e-10: with |u| ~ 200 - values ~ -2e4 on f16..f24 states - the value amplifies the rounding of |v|^2; measured 1.2e-11)
    assert np.abs(vb - vr[:96]).max() <= 1e-10 * max(1.0, np.abs(vb).max()) and s["grad_large"] < 1e-6, s
    assert np.array_equal(gb[:12], g[20:32])                   # both finishing paths sum in the same order
    # ---- finite differences of the device value against the device gradient --------------------------------------
    h = 1e-6
    Xf = rec.trace.cands
    _, gf = ctx.acq_eval(Xf, rec.best_f, False)
    fd_err = 0.0
    for j in sorted({0, k // 2, k - 1}):
        Xp, Xm = Xf.copy(), Xf.copy()
        Xp[:, j] += h
        Xm[:, j] -= h
        fd = (ctx.acq_eval(Xp, rec.best_f, False, grad=False) - ctx.acq_eval(Xm, rec.best_f, False, grad=False)) / (2 * h)
        fd_err = max(fd_err, float(np.abs(fd - gf[:, j]).max() / max(1.0, np.abs(gf[:, j]).max())))
    s["fd"] = fd_err
    assert fd_err < 1e-3, s          # measured <= 6e-5 (truncation error of the difference quotient, h = 1e-6)
    # ---- rows M-N: one optimize_acqf from the oracle's initial conditions, both launch modes ---------------------
    outs = []
    for resident in (1, 0):
        ctx.set_option(native.OPT_RESIDENT, resident)
        outs.append(ctx.optimize_acqf(rec.trace.ics, rec.acq_bounds, rec.best_f))
    ctx.set_option(native.OPT_RESIDENT, 1)
    (cand, vals, info, failed), (cand2, vals2, info2, failed2) = outs
    assert np.array_equal(cand, cand2) and np.array_equal(vals, vals2) and np.array_equal(info, info2)   # bit for bit
    assert failed == failed2
    if not rec.trace.retried:
        assert not failed, n
    lbt = rec.trace.lbfgsb[:len(info)]              # (after a retry the oracle's trace continues with the second attempt)
    s["counts_equal"] = [bool((t.nit, t.nfev) == (int(info[i, 0]), int(info[i, 1]))) for i, t in enumerate(lbt)]
    if not rec.trace.retried:
        s["cand"] = (np.abs(cand - rec.trace.cands).max(axis=1) / max(1.0, np.abs(rec.trace.cands).max())).tolist()
        s["vals"] = (np.abs(vals - rec.trace.vals) / np.maximum(1.0, np.abs(rec.trace.vals))).tolist()
        # the device's surface at the device's own end points, judged by the oracle
        vo = acq(torch.from_numpy(np.ascontiguousarray(cand))).detach().numpy()
        s["surf"] = float((np.abs(vo - vals) / np.maximum(1.0, np.abs(vals))).max())
        assert s["surf"] < 1e-8, s
    # ---- the optimiser itself, isolated from rounding in f/g: REAL scipy L-BFGS-B driven by the DEVICE's own value and
    # gradient (bit-identical f/g on both sides).  Early states: same iteration and evaluation counts, end points to 1e-7.
    # Late states: the two implementations part ways about as often as device and oracle do, although only the order of
    # their dot products differs (scipy's C port sums in BLAS order) - the sensitivity is the problem's, not the surface's
    # (statistic asserted in _summarise)
    from scipy.optimize import minimize
    s["scipy_on_device_surface"] = []
    for gi in range(len(info)):
        ics = rec.trace.ics[5 * gi:5 * gi + 5]
        b = ics.shape[0]
        lo, hi = np.tile(rec.acq_bounds[0], b), np.tile(rec.acq_bounds[1], b)

        def fun(x):
            vv, gg = ctx.acq_eval(x.reshape(b, k), rec.best_f, False)
            return -float(vv.sum()), -gg.reshape(-1)

        r = minimize(fun, np.clip(ics.reshape(-1), lo, hi), jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)),
                     options={"maxiter": 200})
        xe = np.clip(r.x.reshape(b, k), rec.acq_bounds[0], rec.acq_bounds[1])
        # (scipy's C port sums its dot products in BLAS order, lbfgsb.cpp in index order: same path, last bits differ)
        same = (int(r.nit), int(r.nfev)) == (int(info[gi, 0]), int(info[gi, 1])) and \
            np.abs(xe - cand[5 * gi:5 * gi + 5]).max() < 1e-6 * max(1.0, np.abs(xe).max())
        s["scipy_on_device_surface"].append(bool(same))
        s.setdefault("scipy_detail", []).append([int(r.nit), int(r.nfev), int(info[gi, 0]), int(info[gi, 1]),
                                                 float(np.abs(xe - cand[5 * gi:5 * gi + 5]).max()), str(r.message)])

    # ---- rows N-O: arg-max candidate and inverse map --------------------------------------------------------------
    x = ctx.inverse_map(rec.cand_z)
    s["inv"] = _rel(x, rec.cand_x)
    assert s["inv"] < 1e-9, s
    # the device's own free-running iteration at this state, replayed: same picks of the raw samples
    assert sorted(rec.trace.ic_idx.tolist()) == sorted(tr["ic_idx"].tolist()), n
    stats.append(s)
    return s


def _summarise(stats, name):
    counts = [c for s in stats for c in s["counts_equal"]]
    cands = [c for s in stats for c in s.get("cand", [])]
    vals = [c for s in stats for c in s.get("vals", [])]
    print("[late %s] states %d: counts equal %.3f; scipy on the device surface %.3f; end points median %.2e <1e-2 %.3f <1e-5 %.3f; "
          "values median %.2e <1e-3 %.3f" % (name, len(stats), np.mean(counts), np.mean([c for s in stats for c in s["scipy_on_device_surface"]]),
                                             np.median(cands), np.mean(np.array(cands) < 1e-2), np.mean(np.array(cands) < 1e-5),
                                             np.median(vals), np.mean(np.array(vals) < 1e-3)))
    # Device optimiser vs ORACLE optimiser (each on its own f/g, which agree to ~1e-13): L-BFGS-B on 5k joint variables
    # stops on a relative f-reduction of 2.2e-9, i.e. on a flat optimum the end point is fixed to ~1e-4 only, and a
    # line-search branch can flip on a 1e-14 difference (EXPERIMENTS.md section 6).  In the late phase of the headline run
    # (k = 8..16, many penalised points) that happens in about a third of the restart groups - measured on MI355X: counts
    # identical for 62 % (d=40, n = 120..449) / 70 % (d=20), end points: median 8e-7 / 6e-12, values: median 3e-11.  The optimiser itself is pinned
    # exactly above (scipy on the device surface); these bounds only catch a surface that has gone wrong.
    # thresholds = measured (round 3, MI355X; deterministic per build) minus 10 %: counts equal 0.625 (d=40) / 0.70 (d=20);
    # real scipy on the device surface 0.4375 / 0.70; end points median 7.9e-7 / 5.8e-12, within 1e-2 0.94 / 0.84, within
    # 1e-5 0.56 / 0.72; values median 3.5e-11 / 3.7e-12, within 1e-3 0.975 / 0.90
    assert np.mean(counts) >= 0.56, counts
    assert np.mean([c for s in stats for c in s["scipy_on_device_surface"]]) >= 0.39
    assert np.median(cands) < 1e-5 and np.mean(np.array(cands) < 1e-2) >= 0.75 and np.mean(np.array(cands) < 1e-5) >= 0.50, np.sort(cands)[-10:]
    assert np.median(vals) < 1e-9 and np.mean(np.array(vals) < 1e-3) >= 0.81, np.sort(vals)[-10:]


def test_headline_run_late_phase_against_oracle(native, headline_run):
    ctx = native.Context(max_n=450, max_d=40, max_q=512)
    stats = []
    for n in HEADLINE_NS:
        rec, tr = _oracle_step(headline_run, n, 40, 0)
        _check_state(native, ctx, rec, tr, stats)
    ctx.close()
    _summarise(stats, "d40")
    assert {s["n"] for s in stats} == set(HEADLINE_NS)


def test_d20_states_against_oracle(native, d20_run):
    ctx = native.Context(max_n=250, max_d=20, max_q=512)
    stats = []
    for n in D20_NS:
        rec, tr = _oracle_step(d20_run, n, 20, 0)
        _check_state(native, ctx, rec, tr, stats)
    ctx.close()
    _summarise(stats, "d20")


def test_headline_late_iterations_replayed_by_oracle(native, headline_run):
    """The device's OWN free-running iterations of the late phase (every 24th from n = 130 on, 14 iterations) replayed
    by the oracle from the same state: raw-sample picks, restart end points, counts, chosen candidate."""
    from test_gpu_parity import _check_replay, _replay_with_oracle

    class _View:                       # the replay helper walks `trace`; hand it the sampled iterations only
        pass

    v = _View()
    v.x_evals, v.f_evals, v.maximization = headline_run.x_evals, headline_run.f_evals, False
    v.trace = [headline_run.trace[i] for i in range(10, 330, 24)]
    st = _replay_with_oracle(v, lambda: BBOBProblem(15, 0, 40), 40)
    _check_replay(st, min_iters=len(v.trace) - 2, late=True)
