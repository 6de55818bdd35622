"""Host-side control flow of `botorch.optim.optimize_acqf(q=1, num_restarts, raw_samples, batch_limit=5, maxiter=200)`
as both optimisers of the reference call it (PCA_BO.py:607-614, Vanilla_BO.py:206-213), with every numerical step on
the device: raw-sample scoring (`Context.acq_eval`), multi-start L-BFGS-B (`Context.optimize_acqf`).  The RNG-consuming
steps (Sobol scramble bits, multinomial pick) run on torch's global CPU generator in botorch's order."""
from __future__ import annotations

import warnings
from time import perf_counter
from typing import Optional

import numpy as np

from . import _native
from . import initializers as _init


def optimize_acqf(ctx: "_native.Context", bounds: np.ndarray, best_f: float, maximize: bool, acq_code: int,
                  num_restarts: int, raw_samples: int, batch_limit: int = 5, maxiter: int = 200, engine=None,
                  breakdown: Optional[dict] = None, trace: Optional[dict] = None, raw: Optional[np.ndarray] = None,
                  raw_vals: Optional[np.ndarray] = None, before_lbfgsb=None):
    """Returns (candidate[1, k], all restart candidates, their values, L-BFGS-B info).  `engine`: a scrambled Sobol
    engine prepared earlier (same RNG consumption, earlier in time); `raw`: the raw samples already drawn from it
    (while the device was still factorising); `raw_vals`: their acquisition values if the caller already has them
    (`Context.gp_wait_eval`).  The retry path draws and scores afresh.  `before_lbfgsb()`: called once, after the
    initial conditions are picked (where PCA_BO starts the next iteration's noise draw on its worker thread)."""
    pb = breakdown if breakdown is not None else {}
    engines = [engine] if engine is not None and raw is None else []
    ready = [raw] if raw is not None else []
    ready_vals = [raw_vals] if raw is not None and raw_vals is not None else []

    def initial_conditions():
        t0 = perf_counter()
        raw = ready.pop() if ready else _init.draw_sobol(bounds, raw_samples, engines.pop() if engines else None)
        t1 = perf_counter()
        vals = ready_vals.pop() if ready_vals else ctx.acq_eval(raw, best_f, maximize, acq_code, grad=False)
        t2 = perf_counter()
        if acq_code == _native.ACQ_PI:
            idx = _init.initialize_q_batch_nonneg(vals, num_restarts)
        else:
            idx = _init.initialize_q_batch(vals, num_restarts)
        t3 = perf_counter()
        pb["sobol"] = pb.get("sobol", 0.0) + t1 - t0
        pb["raw_eval"] = pb.get("raw_eval", 0.0) + t2 - t1
        pb["init_pick"] = pb.get("init_pick", 0.0) + t3 - t2
        if trace is not None:
            trace.update(raw_vals=vals.copy(), ic_idx=np.asarray(idx).copy())
        return raw[idx]

    ics = initial_conditions()
    if before_lbfgsb is not None:
        before_lbfgsb()
    t_opt = perf_counter()
    cand, vals, info, failed = ctx.optimize_acqf(ics, bounds, best_f, maximize, acq_code, batch_limit=batch_limit,
                                                 maxiter=maxiter)
    pb["lbfgsb"] = pb.get("lbfgsb", 0.0) + perf_counter() - t_opt
    if failed:   # botorch: OptimizationWarning -> one retry with freshly drawn initial conditions
        warnings.warn("Optimization failed in `gen_candidates_scipy`; trying again with a new set of "
                      "initial conditions.", RuntimeWarning)
        if trace is not None:
            trace["retried"] = True
        ics = initial_conditions()
        cand, vals, info, failed = ctx.optimize_acqf(ics, bounds, best_f, maximize, acq_code,
                                                     batch_limit=batch_limit, maxiter=maxiter)
    best = int(np.argmax(vals))
    if trace is not None:
        trace.update(ics=ics.copy(), cands=cand.copy(), vals=vals.copy(), chosen=best, info=info.copy(),
                     k=int(cand.shape[1]))
    return cand[best].reshape(1, -1), cand, vals, info
