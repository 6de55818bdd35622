"""Host-side initial-condition logic of the acquisition optimiser (rows K, L of SURVEY.md 8a).

This is control flow + RNG consumption, not arithmetic: it must draw from the *global torch CPU
generator* in exactly the order botorch does, so that runs are reproducible against the reference
on the same seed (reference call site: Algorithms/BayesianOptimization/PCA_BO.py:607-614 ->
botorch.optim.initializers.gen_batch_initial_conditions).  torch is used here as the reference's
dependency uses it: `SobolEngine(scramble=True, seed=None)` and `torch.multinomial`.
"""
from __future__ import annotations

import math
import warnings

import numpy as np
import torch

INIT_ETA = 1.0   # botorch initialize_q_batch(eta=1.0)


_MAXBIT = 30
_POW_LOW = torch.pow(2, torch.arange(0, _MAXBIT))
_POW_LOW_NP = _POW_LOW.numpy().astype(np.int64)


_UNSCRAMBLED = {}      # k -> (k, 30) int64 direction numbers of torch's unscrambled engine (its table; copied per use)


class ScrambledSobol:
    """State of a fresh `SobolEngine(k, scramble=True)`: the scrambled direction numbers and the shift.  `draw(n)` gives the
    points torch's engine would give on its first draw (bit-identical; the engine is single-use, like the ones botorch
    builds per call)."""
    __slots__ = ("k", "state", "shift")

    def __init__(self, k, state, shift):
        self.k, self.state, self.shift = k, state, shift

    def draw(self, n: int, dtype=torch.float64) -> torch.Tensor:
        from . import _native
        return torch.from_numpy(_native.sobol_draw(self.state, self.shift, n)).to(dtype)


def scrambled_sobol_engine(k: int, generator=None) -> ScrambledSobol:
    """The engine `SobolEngine(k, scramble=True, seed=None)` would be, ~4x faster to construct.
    `generator`: a `torch.Generator` standing in for torch's global CPU generator (same stream for the same seed) - used
    where several runs share one process (pcabo.batchrun); None = the global generator, as botorch.

    The two `torch.randint` draws are exactly the ones torch makes (same consumption of the CPU generator, same order);
    the matrix scramble (torch's `_sobol_engine_scramble_`, a scalar accessor loop) and the draw run in libpcabo's host
    helpers `pcabo_sobol_scramble` / `pcabo_sobol_draw`.  Pinned against the real engine in tests/test_abi_and_host.py."""
    from . import _native
    base = _UNSCRAMBLED.get(k)
    if base is None:
        base = _UNSCRAMBLED[k] = torch.quasirandom.SobolEngine(k, scramble=False).sobolstate.numpy().copy()
    from .hostrng import HostMT
    if isinstance(generator, HostMT):       # the run's generator as a state blob: the same draws without torch calls (pcabo/hostrng.py)
        shift = generator.randint2((k, _MAXBIT)) @ _POW_LOW_NP
        ltm = generator.randint2((k, _MAXBIT, _MAXBIT))
    else:
        shift_ints = torch.randint(2, (k, _MAXBIT), generator=generator)
        shift = torch.mv(shift_ints, _POW_LOW).numpy()
        ltm = torch.randint(2, (k, _MAXBIT, _MAXBIT), generator=generator).numpy()     # (torch's .tril(): the helper reads below the diagonal only)
    state = base.copy()
    _native.sobol_scramble(state, ltm)
    return ScrambledSobol(k, state, shift)


def draw_sobol(bounds: np.ndarray, n: int, engine=None, out=None) -> np.ndarray:
    """botorch `draw_sobol_samples(bounds, n, q=1, seed=None)` -> n x k points inside `bounds` (2 x k).
    `engine`: a fresh engine from `scrambled_sobol_engine(k)` built earlier (same RNG consumption, earlier in time)."""
    from . import _native
    k = bounds.shape[1]
    if engine is None:
        engine = scrambled_sobol_engine(k)
    return _native.sobol_draw(engine.state, engine.shift, n, bounds[0], bounds[1] - bounds[0], out=out)


def _with_torch_generator(fn):
    """`generator` may be a pcabo.hostrng.HostMT: the call runs on a real torch generator in its state, which is taken back after."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, generator=None, **kw):
        from .hostrng import HostMT
        if isinstance(generator, HostMT):
            g = generator.torch_generator()
            try:
                return fn(*args, generator=g, **kw)
            finally:
                generator.absorb(g)
        return fn(*args, generator=generator, **kw)
    return wrapped


@_with_torch_generator
@torch.inference_mode()
def initialize_q_batch(acq_vals: np.ndarray, n: int, eta: float = INIT_ETA, generator=None) -> np.ndarray:
    """Boltzmann sampling of n restart indices (without replacement) + forced arg-max."""
    v = torch.from_numpy(np.ascontiguousarray(acq_vals, dtype=np.float64))
    n_samples = v.shape[0]
    if n > n_samples:
        raise RuntimeError(f"n ({n}) cannot be larger than the number of provided samples ({n_samples})")
    if n == n_samples:
        return np.arange(n)
    std = v.std(dim=0)
    if float(std) == 0.0:
        warnings.warn("All acquisition values for raw samples points are the same. "
                      "Choosing initial conditions at random.", RuntimeWarning)
        return torch.randperm(n=n_samples, generator=generator)[:n].numpy()
    max_idx = int(torch.max(v, dim=0)[1])
    eta_z = eta * ((v - v.mean(dim=0)) / std)
    weights = torch.exp(eta_z)
    # botorch halves eta_z while exp overflows.  |z| <= (n - 1) / sqrt(n) for a sample of n with the unbiased std, so for
    # eta (n - 1) / sqrt(n) < 700 the check cannot fire and its two tensor ops are skipped
    if eta * (n_samples - 1) / math.sqrt(n_samples) >= 700.0:
        while bool(torch.isinf(weights).any()):
            eta_z = eta_z * 0.5
            weights = torch.exp(eta_z)
    idcs = torch.multinomial(weights, n, generator=generator).numpy()
    if max_idx not in idcs:
        idcs[-1] = max_idx
    return idcs


def initialize_q_batch_rows(acq_vals: np.ndarray, n: int, generators, eta: float = INIT_ETA, skip=None):
    """initialize_q_batch for every row of acq_vals (B x n_samples, one run per row, its own generator): the statistics that are
    bit-identical row-wise and on the 2-D tensor (mean, arg-max, the element-wise weights) are formed for all rows at once, the
    standard deviation - torch's 2-D reduction rounds it differently - and the multinomial draw per row.  Returns what B calls
    of initialize_q_batch return (rows listed in `skip` get arange(n) without touching their generator)."""
    v = torch.from_numpy(np.ascontiguousarray(acq_vals, dtype=np.float64))
    B, n_samples = v.shape
    skip = set(skip or ())
    if n > n_samples:
        raise RuntimeError(f"n ({n}) cannot be larger than the number of provided samples ({n_samples})")
    if n == n_samples:
        return [np.arange(n) for _ in range(B)]
    from .hostrng import boltzmann_pick_rows
    native = boltzmann_pick_rows(acq_vals, n, eta, generators, skip) if n_samples >= 2 else None
    if native is not None:
        # every run's generator is a state blob: statistics, weights, multinomial draw and the forced arg-max of all rows in one
        # call into libpcabo (csrc/host_entry.cpp; the same picks as the torch path below - tests/test_abi_and_host.py)
        idx, flags = native
        out = []
        for b in range(B):
            if flags[b] == 2:
                out.append(np.arange(n))
            elif flags[b] == 1:
                warnings.warn("All acquisition values for raw samples points are the same. "
                              "Choosing initial conditions at random.", RuntimeWarning)
                tg = generators[b].torch_generator()
                out.append(torch.randperm(n=n_samples, generator=tg)[:n].numpy())
                generators[b].absorb(tg)
            else:
                out.append(idx[b])
        return out
    std = torch.stack([v[b].std(dim=0) for b in range(B)])
    mean = v.mean(dim=1)
    max_idx = torch.max(v, dim=1)[1]
    safe = torch.where(std == 0.0, torch.ones_like(std), std)
    weights = torch.exp(eta * ((v - mean[:, None]) / safe[:, None]))
    check = eta * (n_samples - 1) / math.sqrt(n_samples) >= 700.0
    from .hostrng import HostMT, multinomial_rows
    std_np = std.numpy()
    out = [None] * B
    draw = []
    for b in range(B):
        if b in skip:
            out[b] = np.arange(n)
        elif std_np[b] == 0.0:
            warnings.warn("All acquisition values for raw samples points are the same. "
                          "Choosing initial conditions at random.", RuntimeWarning)
            g = generators[b]
            tg = g.torch_generator() if isinstance(g, HostMT) else g
            out[b] = torch.randperm(n=n_samples, generator=tg)[:n].numpy()
            if isinstance(g, HostMT):
                g.absorb(tg)
        else:
            draw.append(b)
    if check:
        for b in draw:
            w, eta_z = weights[b], eta * ((v[b] - mean[b]) / std[b])
            while bool(torch.isinf(w).any()):
                eta_z = eta_z * 0.5
                w = torch.exp(eta_z)
            weights[b] = w
    # the multinomial draws of all rows: one call into libpcabo for the runs whose generator is a state blob (pcabo/hostrng.py:
    # torch's exponential variates and top-k restated, checked against torch at import), torch's own call for the others
    picks = multinomial_rows(weights.numpy(), n, generators, draw)
    mi = max_idx.numpy()
    for b in draw:
        idcs = picks[b]
        if mi[b] not in idcs:
            idcs[-1] = mi[b]
        out[b] = idcs
    return out


@_with_torch_generator
def initialize_q_batch_nonneg(acq_vals: np.ndarray, n: int, eta: float = 1.0, alpha: float = 1e-4,
                              generator=None) -> np.ndarray:
    """Variant botorch uses for non-negative acquisitions (probability of improvement)."""
    v = torch.from_numpy(np.ascontiguousarray(acq_vals, dtype=np.float64))
    n_samples = v.shape[0]
    if n == n_samples:
        return np.arange(n)
    max_val, max_idx = torch.max(v, dim=0)
    if bool(max_val <= 0):
        warnings.warn("All acquisition values for raw sampled points are nonpositive, so initial conditions "
                      "are being selected randomly.", RuntimeWarning)
        return torch.randperm(n=n_samples, generator=generator)[:n].numpy()
    pos = v > 0
    num_pos = int(pos.sum())
    if num_pos < n:
        remaining = (~pos).nonzero(as_tuple=False).view(-1)
        rand = torch.randperm(remaining.shape[0], generator=generator)
        pos[remaining[rand[: n - num_pos]]] = 1
        return pos.nonzero(as_tuple=False).view(-1).numpy()
    alpha_pos = v >= alpha * max_val
    while int(alpha_pos.sum()) < n:
        alpha = 0.1 * alpha
        alpha_pos = v >= alpha * max_val
    alpha_pos_idcs = torch.arange(len(v))[alpha_pos]
    weights = torch.exp(eta * (v[alpha_pos] / max_val - 1))
    idcs = alpha_pos_idcs[torch.multinomial(weights, n, generator=generator)]
    if max_idx not in idcs:
        idcs[-1] = max_idx
    return idcs.numpy()
