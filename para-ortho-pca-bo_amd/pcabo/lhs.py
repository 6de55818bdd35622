"""Latin-hypercube designs as the reference draws them (pyDOE 0.3.8 `lhs`, all four criteria its LHS_sampler accepts).

The reference calls pyDOE 0.3.8 `lhs(n=dim, samples=m, criterion="center", iterations=...)`
(reference: Algorithms/BayesianOptimization/AbstractBayesianOptimizer.py:40-45), which is not
installed here.  pyDOE's centred variant consumes the *legacy global* numpy RNG as
`rand(m, dim)` (drawn, unused) followed by one `permutation` of the bin centres per column.
Pinned by the first n_DoE rows of all 120 runs in the reference's committed .dat files
(tests/golden/doe_f15_f20_dim5.npz).

"maximin", "centermaximin" and "correlation" (accepted by the reference's `LHS_sampler`,
AbstractBayesianOptimizer.py:8-103; the default of a bare `LHS_sampler()` is "correlation", the reference's runner and
example scripts always pass "center") are restated from pyDOE 0.3.8's published algorithm: `iterations` candidate designs,
the best by the criterion kept.  **Parity unpinned**: the reference commits no run drawn with them; what is asserted
(tests/test_abi_and_host.py) is the structure - Latin-hypercube property, RNG consumption per candidate (one `rand(m, dim)`
block, then one permutation per column), the criterion's monotone improvement over a single candidate.
"""
from __future__ import annotations

import numpy as np

__all__ = ["lhs_center", "lhs"]


def lhs_center(dim: int, samples: int, rng=None) -> np.ndarray:
    """`rng`: a `np.random.RandomState` standing in for the legacy global generator (same stream for the same seed) -
    used where several runs share one process (pcabo.batchrun); None = the global generator, as the reference."""
    rng = np.random if rng is None else rng
    cut = np.linspace(0, 1, samples + 1)
    rng.rand(samples, dim)                    # pyDOE draws this and then ignores it for "center"
    centres = (cut[:samples] + cut[1:samples + 1]) / 2
    h = np.empty((samples, dim))
    for j in range(dim):
        h[:, j] = rng.permutation(centres)
    return h


def _lhs_classic(dim: int, samples: int, rng) -> np.ndarray:
    """pyDOE `_lhsclassic`: a uniform point in every bin, the bins of a column in random order."""
    cut = np.linspace(0, 1, samples + 1)
    u = rng.rand(samples, dim)
    a, b = cut[:samples], cut[1:samples + 1]
    rd = u * (b - a)[:, None] + a[:, None]
    h = np.empty_like(rd)
    for j in range(dim):
        order = rng.permutation(samples)
        h[:, j] = rd[order, j]
    return h


def _pdist_min(x: np.ndarray) -> float:
    """Smallest pairwise Euclidean distance of the rows (pyDOE `_pdist`, its minimum)."""
    g = x @ x.T
    sq = np.diag(g)[:, None] + np.diag(g)[None, :] - 2.0 * g
    iu = np.triu_indices(x.shape[0], 1)
    return float(np.sqrt(np.maximum(sq[iu], 0.0)).min()) if len(iu[0]) else 0.0


def lhs(dim: int, samples: int, criterion: str = "center", iterations: int = 5, rng=None) -> np.ndarray:
    """pyDOE 0.3.8 `lhs(n=dim, samples=samples, criterion=..., iterations=...)` on the legacy numpy RNG (`rng`: a RandomState
    standing in for the global generator).  criterion: "center", "maximin", "centermaximin", "correlation"."""
    rng = np.random if rng is None else rng
    if criterion == "center":
        return lhs_center(dim, samples, rng)
    if criterion in ("maximin", "centermaximin"):
        best, maxdist, cand = None, 0.0, None
        for _ in range(int(iterations)):
            cand = _lhs_classic(dim, samples, rng) if criterion == "maximin" else lhs_center(dim, samples, rng)
            d = _pdist_min(cand)
            if maxdist < d:
                maxdist, best = d, cand.copy()
        return best if best is not None else cand    # (no candidate with a positive distance: one point, or coincident rows)
    if criterion == "correlation":
        best, mincorr = None, np.inf
        for _ in range(int(iterations)):
            cand = _lhs_classic(dim, samples, rng)
            r = np.corrcoef(cand)                    # (pyDOE correlates the ROWS - kept)
            off = np.abs(r - np.eye(r.shape[0]))
            m = float(np.max(np.abs(r[r != 1]))) if np.any(r != 1) else 0.0
            if m < mincorr:
                mincorr = float(np.max(off))
                best = cand.copy()
        return best
    raise ValueError("The criterion is not matching the set ones!")
