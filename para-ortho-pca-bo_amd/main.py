#!/usr/bin/env python3
"""Command line entry for comparison experiments (same flags as the reference's main.py:16-94).

    python main.py --quick                                   # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --dimensions 40 --problems 15
                                                             # runs sharded over 8 GPUs, one process per GPU
"""
import argparse
import os
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # read by the HIP runtime at its first call: a Batch uses a stream per worker thread

from Algorithms import ExperimentRunner


def parse_arguments():
    p = argparse.ArgumentParser(description="Run Bayesian Optimization comparison experiments on MI355X.")
    p.add_argument("--dimensions", type=int, nargs="+", default=[10, 20, 40])
    p.add_argument("--problems", type=int, nargs="+", default=[15, 16, 17])
    p.add_argument("--runs", type=int, default=30)
    p.add_argument("--budget_factor", type=int, default=10, help="budget = budget_factor * dim + 50")
    p.add_argument("--doe_factor", type=int, default=3, help="n_doe = doe_factor * dim")
    p.add_argument("--experiment_dir", type=str, default="experiment")
    p.add_argument("--acquisition", type=str, default="expected_improvement",
                   choices=["expected_improvement", "probability_of_improvement", "upper_confidence_bound"])
    p.add_argument("--var_threshold", type=float, default=0.95)
    p.add_argument("--algorithms", type=str, nargs="+", default=["pca", "vanilla"], choices=["pca", "vanilla"])
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--quick", action="store_true", help="5-D, f15 + f20, budget factor 5, DoE factor 2")
    # not in the reference: PCA_BO runs of one dimension advance in lock-step, `--batched` at a time (0: one run after the other)
    p.add_argument("--batched", type=int, default=0, help="PCA_BO runs per lock-step batch (same runs, same numbers)")
    p.add_argument("--side_by_side", type=int, default=2, help="lock-step batches advancing at once (one host thread each)")
    p.add_argument("--batch_acq_kernel", default="group", choices=["group", "latency", "device", "auto"],
                   help="'device': every restart group's L-BFGS-B inside one kernel launch, the batches interleaved on one host "
                        "thread - for many runs per GPU (e.g. --batched 75 --side_by_side 4); 'auto': 'device' for a dimension "
                        "20 <= d <= 40 with 30 or more runs on this GPU, 'group' otherwise")
    return p.parse_args()


def main():
    a = parse_arguments()
    if a.quick:                                   # the reference's quick configuration (main.py:103-109)
        a.dimensions, a.problems, a.runs, a.budget_factor, a.doe_factor = [5], [15, 20], 30, 5, 2.0
    rank = int(os.environ.get("RANK", "0"))
    experiment = ExperimentRunner(
        algorithms=a.algorithms, dimensions=a.dimensions, problem_ids=a.problems, num_runs=a.runs,
        budget_factor=a.budget_factor, doe_factor=a.doe_factor, root_dir=os.getcwd(), experiment_name=a.experiment_dir,
        acquisition_function=a.acquisition, pca_components=0, var_threshold=a.var_threshold, verbose=a.verbose,
        progress=(rank == 0), batched=a.batched, side_by_side=a.side_by_side, batch_acq_kernel=a.batch_acq_kernel)
    t0 = time.time()
    experiment.run_experiment()
    dt = time.time() - t0
    its = sum(r["iterations"] for r in experiment.results)
    print(f"[rank {rank}] {len(experiment.results)} runs, {its} BO iterations in {dt:.2f} s ({its / dt:.1f} it/s)")


if __name__ == "__main__":
    main()
