"""Host-side shortcuts of the lock-step driver (pcabo/batchrun.py) must give, bit for bit, what the per-run calls give -
they feed RNG draws and rank ties that decide the runs' paths (reference: PCA_BO.py:330-333 ranks; botorch
initialize_q_batch behind gen_batch_initial_conditions)."""
import warnings

import numpy as np
import torch

from pcabo import initializers as I


def test_boltzmann_picks_of_all_runs_equal_the_per_run_calls():
    """initialize_q_batch_rows: mean / arg-max / weights from one 2-D expression, standard deviation and multinomial per run -
    same indices and the same generator state afterwards, including the all-equal row (random permutation) and skipped rows."""
    rng = np.random.default_rng(1)
    for t in range(60):
        B = 9
        vals = rng.normal(size=(B, 512)) * rng.uniform(0.1, 30) + rng.normal() * 5
        if t % 5 == 0:
            vals[3] = 1.234                                   # std == 0: warning + randperm
        if t % 7 == 0:
            vals[:, :64] = vals[:, 64:128]                    # repeated values
        g1 = [torch.Generator().manual_seed(100 + b) for b in range(B)]
        g2 = [torch.Generator().manual_seed(100 + b) for b in range(B)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = [I.initialize_q_batch(vals[b], 10, generator=g1[b]) if b != 5 else np.arange(10) for b in range(B)]
            r = I.initialize_q_batch_rows(vals, 10, g2, skip=[5])
        for b in range(B):
            assert np.array_equal(a[b], r[b]), (t, b)
            assert torch.equal(g1[b].get_state(), g2[b].get_state()), (t, b)


def test_torch_std_is_why_the_standard_deviation_stays_per_run():
    """The reason for the per-run std in initialize_q_batch_rows, pinned: torch's reduction over the rows of a 2-D tensor rounds
    the standard deviation differently from the 1-D call (the mean and the arg-max agree)."""
    rng = np.random.default_rng(0)
    v = torch.from_numpy(rng.normal(size=(30, 512)) * 7.0)
    assert all(v[b].mean(dim=0).item() == v.mean(dim=1)[b].item() for b in range(30))
    assert all(int(torch.max(v[b], dim=0)[1]) == int(torch.max(v, dim=1)[1][b]) for b in range(30))
    # (not asserted to differ - a future torch may change it - only that the shortcut never relied on it)


def test_row_wise_argsort_gives_the_ranks_of_the_1d_calls():
    """Ranks of all runs from argsort along the rows of the B x n array: the same permutation as numpy's 1-D argsort of every row,
    ties (the repeated out-of-box penalty 1000.0) included."""
    rng = np.random.default_rng(3)
    for t in range(40):
        n = int(rng.integers(30, 450))
        F = rng.normal(size=(24, n)) * 100
        F[:, rng.integers(0, n, size=n // 6)] = 1000.0
        if t % 2:
            F[3, :] = 1000.0
        for sign in (1.0, -1.0):
            r2 = np.argsort(np.argsort(sign * F, axis=1), axis=1) + 1
            for b in range(F.shape[0]):
                fb = (sign * F[b]).copy()
                assert np.array_equal(np.argsort(np.argsort(fb)) + 1, r2[b]), (t, b)


def test_device_groups_of_different_dimensions_merge_up_to_eight_batches():
    """ExperimentRunner._run_batched: the device-mode groups of the dimensions advance together while they fit
    DEVICE_BATCHES_AT_ONCE batches; order kept, no batch split, nothing lost."""
    from Algorithms.Experiment.ExperimentRunner import merge_batch_groups, DEVICE_BATCHES_AT_ONCE
    assert DEVICE_BATCHES_AT_ONCE == 8
    a, b, c = [("d20", i) for i in range(4)], [("d40", i) for i in range(4)], [("d10", i) for i in range(3)]
    assert merge_batch_groups([a, b], 8) == [a + b]                        # configs[3]: 4 x 75 + 4 x 75 in one group
    assert merge_batch_groups([a, b, c], 8) == [a + b, c]
    assert merge_batch_groups([c, a, b], 8) == [c + a, b]
    assert merge_batch_groups([a, b], 4) == [a, b] and merge_batch_groups([], 8) == []
    nine = [("x", i) for i in range(9)]
    assert merge_batch_groups([nine, a], 8) == [nine, a]                   # (a group is never split)


def test_device_batch_plan_covers_every_run_in_bounded_batches():
    """ExperimentRunner's "auto" mode: up to four batches of >= 30 runs at a time, none above 120 runs, nothing left over."""
    from Algorithms.Experiment.ExperimentRunner import device_batch_plan
    assert device_batch_plan(45) == (45, 1) and device_batch_plan(90) == (30, 3) and device_batch_plan(600) == (75, 4)
    for n in range(40, 1500, 7):
        per, nb = device_batch_plan(n)
        parts = [min(per, n - i) for i in range(0, n, per)]
        assert sum(parts) == n and max(parts) <= 120 and 1 <= nb <= 4
        assert min(parts) >= min(30, n) // 2          # (the last batch may be short, never a sliver)
