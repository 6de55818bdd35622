"""K(X,X) + Cholesky micro-benchmark on SURVEY.md 8(d)'s grid: Z ~ U[1/12, 11/12]^(n x k), y ~ N(0, 1), fixed seeds,
(n, k) in {(150,10), (250,19), (450,36), (1050,89)} x batch in {1, 30}; device time per phase from HIP events on the
batch's stream (pcabo_batch_get_profile), algorithmic work per SURVEY.md 8(d):
    Gram 2 n^2 k + 12 n^2 flop, 8 n k + 8 n^2 B;  Cholesky n^3/3 flop, 16 n^2 B;  root inverse n^3/3 + 2 n^2 flop, 16 n^2 B.
The points go in as X with n_components = k forced, so that the weighted PCA in front (a rotation) hands the Gram
kernel k-dimensional inputs of exactly this distribution's shape."""
from __future__ import annotations

import numpy as np

from . import _native

GRID = ((150, 10), (250, 19), (450, 36), (1050, 89))
FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0


def run(device: int = 0, batches=(1, 30), grid=GRID, reps: int = 5, big_batch: int = 0) -> list:
    """`big_batch` > 0: the two larger shapes also with that many runs side by side (the regime in which the matrix cores fill)."""
    out = []
    for n, k in grid:
        for B in tuple(batches) + ((big_batch,) if big_batch and n >= 450 else ()):
            rng = np.random.default_rng(1000 * n + B)
            X = rng.uniform(1.0 / 12, 11.0 / 12, (B, n, k))
            y = rng.normal(size=(B, n))
            ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
            bt = _native.Batch(B, max_n=n, max_d=k, max_q=64, device=device)
            bt.set_profiling(True)
            acc = None
            for r in range(reps + 1):
                bt.wpca_gp_condition_begin(X, ranks, None, y, n_components=k)
                bt.wpca_results()
                boxes = bt.acq_bounds()
                q = np.stack([boxes[b].mean(axis=0) for b in range(B)])
                _, status = bt.gp_wait_eval([q[b].reshape(1, -1).repeat(64, 0) for b in range(B)], [float(y[b].min()) for b in range(B)])
                assert not status.any()
                p = bt.condition_profile()
                if r > 0:                       # first pass: warm-up
                    acc = p if acc is None else {key: acc[key] + p[key] for key in p}
            bt.close()
            ms = {key: v / reps for key, v in acc.items()}
            fl_g, by_g = B * (2.0 * n * n * k + 12.0 * n * n), B * (8.0 * n * k + 8.0 * n * n)
            fl_c, by_c = B * (n ** 3 / 3.0), B * 16.0 * n * n
            fl_r = B * (n ** 3 / 3.0 + 2.0 * n * n)
            sec = (ms["gram"] + ms["cholesky"]) * 1e-3
            tf = (fl_g + fl_c) / sec / 1e12
            out.append({"n": n, "k": k, "batch": B, "us": {key: 1e3 * v for key, v in ms.items()},
                        "gram_tflops": fl_g / (ms["gram"] * 1e-3) / 1e12, "gram_GBs": by_g / (ms["gram"] * 1e-3) / 1e9,
                        "cholesky_tflops": fl_c / (ms["cholesky"] * 1e-3) / 1e12,
                        "root_inverse_tflops": fl_r / (ms["root_inverse_alpha"] * 1e-3) / 1e12,
                        "kchol_tflops": tf, "kchol_frac_of_fp64_peak": tf / FP64_PEAK_TFLOPS,
                        "kchol_GBs": (by_g + by_c) / sec / 1e9, "kchol_frac_of_hbm_peak": (by_g + by_c) / sec / 1e9 / HBM_PEAK_GBS})
    return out
