"""Latin-hypercube design, `criterion="center"`, as the reference draws it.

The reference calls pyDOE 0.3.8 `lhs(n=dim, samples=m, criterion="center", iterations=...)`
(reference: Algorithms/BayesianOptimization/AbstractBayesianOptimizer.py:40-45), which is not
installed here.  pyDOE's centred variant consumes the *legacy global* numpy RNG as
`rand(m, dim)` (drawn, unused) followed by one `permutation` of the bin centres per column.
Pinned by the first n_DoE rows of all 120 runs in the reference's committed .dat files
(tests/golden/doe_f15_f20_dim5.npz).
"""
from __future__ import annotations

import numpy as np

__all__ = ["lhs_center"]


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
