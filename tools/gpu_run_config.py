"""Run one of BASELINE.json's multi-run configurations end to end on ONE GPU through the product's own runner
(`ExperimentRunner(batched=30)`: the 30 instances of a (function, dimension) cell advance in lock-step) and print one
JSON line: runs, BO iterations, wall seconds, aggregate BO iterations/s, per-dimension breakdown, IOHprofiler files written.
    python tools/gpu_run_config.py 2 [batched] [side_by_side] [acq_kernel]     # configs[2]: f15/f16/f17 x d in {10, 20, 40} x 30 runs
    python tools/gpu_run_config.py 3 75 4 auto together     # one runner call for all dimensions
    python tools/gpu_run_config.py 3      # configs[3] on one GPU: f15-f24 x d in {20, 40} x 30 runs (the N = 1 point)"""
import json, os, sys, tempfile, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # read by the HIP runtime at its first call: a Batch uses a stream per worker thread
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import torch
from Algorithms import ExperimentRunner
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batched = int(sys.argv[2]) if len(sys.argv) > 2 else 30          # runs per lock-step batch
side_by_side = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # batches advancing at once
acq_kernel = sys.argv[4] if len(sys.argv) > 4 else "group"       # "device": device-resident L-BFGS-B, the batches interleaved on one host thread
fids, dims = ([15, 16, 17], [10, 20, 40]) if which == 2 else (list(range(15, 25)), [20, 40])
torch.set_num_threads(4)
root = tempfile.mkdtemp(prefix="pcabo_cfg%d_" % which)
per_dim = {}
t_all = time.perf_counter()
together = len(sys.argv) > 5 and sys.argv[5] == "together"     # ONE runner call for all dimensions (the product's way: device-mode batches of different dimensions advance together)
for dim in ([dims] if together else dims):                       # else one runner call per dimension so that each gets its own clock
    er = ExperimentRunner(algorithms=["pca"], dimensions=dim if together else [dim], problem_ids=fids, num_runs=30, root_dir=root,
                          experiment_name="experiment-all" if together else f"experiment-d{dim}", progress=False, batched=batched, side_by_side=side_by_side, batch_acq_kernel=acq_kernel)
    t0 = time.perf_counter()
    er.run_experiment()
    dt = time.perf_counter() - t0
    its = sum(r["iterations"] for r in er.results)
    per_dim["all" if together else dim] = {"runs": len(er.results), "bo_iterations": its, "seconds": dt, "bo_iterations_per_s": its / dt,
                    "runs_stopped_early": [(f["problem_id"], f["instance"], f["n"]) for f in er.failed_runs],
                    "best_by_function": {str(f): min(r["best"] for r in er.results if r["problem_id"] == f) for f in fids}}
    print(f"d={dim}: {len(er.results)} runs, {its} BO iterations in {dt:.1f} s = {its / dt:.0f} it/s", file=sys.stderr, flush=True)
total = time.perf_counter() - t_all
files = sum(len(f) for _, _, f in os.walk(root))
its = sum(v["bo_iterations"] for v in per_dim.values())
print(json.dumps({"config": f"BASELINE.json configs[{which}] on one MI355X: PCA_BO, functions {fids}, dimensions {dims}, 30 instances each, "
                            f"ExperimentRunner(batched={batched}, side_by_side={side_by_side}, batch_acq_kernel={acq_kernel!r})", "runs": sum(v["runs"] for v in per_dim.values()), "bo_iterations": its,
                  "seconds": total, "aggregate_bo_iterations_per_s": its / total, "per_dimension": per_dim, "files_written": files}))
