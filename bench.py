#!/usr/bin/env python
"""Benchmark of the PCA_BO inner loop on MI355X:  BO iterations / second on BBOB f15, d = 40.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one BO iteration (rank-weighted PCA -> GP re-conditioning -> 512 raw samples + 10-restart
L-BFGS-B over log-EI -> inverse map -> objective) of the configuration BASELINE.json quotes the metric
on: configs[1] = PCA_BO on BBOB f15, d=40, budget 450, n_DoE 120 (330 BO iterations, n grows 120 -> 449).

The cost of an iteration grows with n (2.3 ms at n=120, ~3.6 ms at n=449), so the K timed steps are SPREAD EVENLY OVER
THE WHOLE RUN: iteration indices floor((i + 1/2) * 330 / K); the iterations in between are advanced untimed (a whole
run costs about a second).  Each timed step is bracketed by a device synchronisation, the job by barriers; the time
of a rank is the sum of its K step times, the job's time the maximum over ranks.  With K >= 330 every iteration of the
run is timed (and further runs of the rank's list follow, their DoE and context set-up untimed).

N > 1: the job is the run list {f15, d=40, instances 0..N-1} - one run per GPU, fixed work per GPU (weak scaling) -
partitioned with the product's own `pcabo.sharding.assign_runs`; no data-path collective; the best-so-far values are
gathered once after the timed region (RCCL; the backend actually used is recorded in the line).

Output: ONE JSON line on rank 0 with `roofline` (dominant kernel), `roofline_kchol` (K(X,X) + Cholesky, north_star's
named step), `cpu_baseline` (N = 1 only) and, with --batch B, the aggregate rate of B runs advancing together.
Data: synthetic (in-repo BBOB f15 restatement, pinned by the reference's own known answers).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # read by the HIP runtime at its first call: a Batch uses a stream per worker thread

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pcabo import distributed as D  # noqa: E402
from pcabo import sharding  # noqa: E402
from pcabo.bbob import BBOBProblem  # noqa: E402

FID, DIM, BUDGET, NDOE = 15, 40, 450, 120
ITERS = BUDGET - NDOE
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = matrix peak (spec)


def spread(k: int, total: int = ITERS):
    """k iteration indices spread evenly over [0, total) (all of them when k >= total)."""
    if k >= total:
        return list(range(total))
    return sorted({int((i + 0.5) * total / k) for i in range(k)})


class Run:
    """One PCA_BO run of the list, advanced iteration by iteration."""

    def __init__(self, device: int, run):
        from Algorithms import PCA_BO
        fid, dim, inst = run
        st = sharding.run_settings(run)
        self.run, self.problem = run, BBOBProblem(fid, inst, dim)
        self.opt = PCA_BO(budget=st["budget"], n_DoE=st["n_doe"], var_threshold=0.95,
                          acquisition_function="expected_improvement", random_seed=st["seed"], maximization=False,
                          verbose=False, device=device, DoE_parameters={"criterion": "center", "iterations": 1000})
        self.opt._start(self.problem)             # seeding + DoE (n_DoE objective calls) + device context: set-up
        self.iteration = 0

    @property
    def n(self):
        return len(self.opt.f_evals)

    def step(self):
        self.opt._bo_iteration(self.problem)
        self.iteration += 1

    def close(self):
        out = {"best": float(self.opt.current_best), "total_times": dict(self.opt.total_times),
               "phase": dict(self.opt.phase_breakdown),
               "rounds": int(sum(int(i[:, 1].max()) for i in self.opt.lbfgsb_info))}
        self.opt._finish()
        return out


def cpu_baseline(states, threads: int):
    """Time the CPU oracle (restated reference path) on a bounded sample: one BO iteration from each of the
    sampled states of the SAME run (teacher-forced), spread over n = 120..449."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcabo_oracle as O
    torch.set_num_threads(threads)
    times, ns = [], []
    for X, f in states:
        prob = BBOBProblem(FID, 0, DIM)
        orc = O.OraclePCABO(budget=BUDGET, n_DoE=NDOE, random_seed=15400)
        orc.x_evals = [row.copy() for row in X]
        orc.f_evals = [float(v) for v in f]
        orc._assign_new_best()
        np.random.seed(7)
        torch.manual_seed(7)
        t0 = time.perf_counter()
        orc.step(prob, np.full(DIM, -5.0), np.full(DIM, 5.0))
        times.append(time.perf_counter() - t0)
        ns.append(len(f))
    mean_t = float(np.mean(times))
    return {"value": 1.0 / mean_t, "unit": "BO iterations/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} teacher-forced BO iterations of the d=40 run at n={ns} (mean {mean_t:.3f} s/iteration); "
                      "oracle = numpy/sklearn/torch-fp64-autograd/scipy L-BFGS-B restatement of the reference path "
                      f"with BoTorch's call granularity; torch threads = {threads} (fastest of 1..32 on this host); "
                      f"botorch importable: {_has('botorch')}"}


def _has(mod: str) -> bool:
    import importlib.util
    return importlib.util.find_spec(mod) is not None


def profiled_pass(device: int, timed_at):
    """Second pass over the same run: the iterations in `timed_at` run with HIP-event profiling on the context's stream
    (one plain launch per L-BFGS-B evaluation), everything in between advances unprofiled."""
    r = Run(device, (FID, DIM, 0))
    ctx = r.opt.device_context
    ctx.reset_profile()
    at = set(timed_at)
    ns = []
    for it in range(ITERS):
        if it in at:
            ns.append(r.n)
            ctx.set_profiling(True)
            r.step()
            ctx.set_profiling(False)
            if it == max(at):
                break
        else:
            r.step()
    prof = ctx.profile()
    calib = ctx.profile_calibration()
    r.close()
    return prof, ns, calib


def rocprof_summary():
    """Per-kernel averages of the committed rocprofv3 --kernel-trace --stats run of THIS command (tools/gpu_profile.sh writes
    it, profiles/<round>/bench_steps20_kernel_stats.csv): the plain-launch acquisition kernel (one launch per round, what the
    profiled pass executes) and the resident one (what the timed region executes, one launch per optimize call)."""
    import csv
    for rel in ("r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", rel, "bench_steps20_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        plain = resident = (0, 0.0)
        for row in csv.DictReader(open(path)):
            name = row["Name"]
            if "k_acq_fast<" not in name:
                continue
            calls, tot = int(row["Calls"]), float(row["TotalDurationNs"])
            if ", false>" in name.split("(")[0]:
                plain = (plain[0] + calls, plain[1] + tot)
            elif ", true>" in name.split("(")[0]:
                resident = (resident[0] + calls, resident[1] + tot)
        return {"source": f"profiles/{rel}/bench_steps20_kernel_stats.csv",
                "plain_launches": plain[0], "plain_avg_us": plain[1] / plain[0] / 1e3 if plain[0] else None,
                "resident_launches": resident[0], "resident_avg_us_per_call": resident[1] / resident[0] / 1e3 if resident[0] else None}
    return None


def batch_per_gpu(device: int, rank: int, size: int, B: int = 30):
    """N > 1: what the product ships per GPU.  Every rank runs ITS SHARE OF configs[3] (30 runs x f15..f24 x d in {20, 40} = 600
    runs, partitioned by the runner's own pcabo.sharding.assign_runs) through `ExperimentRunner(batched=75, side_by_side=4,
    batch_acq_kernel="auto")` - the plan main.py runs: `auto` resolves to the device-resident optimiser for both dimensions from
    the EXPERIMENT (300 runs per dimension), whatever the rank's share, so the rows are those of the 1-GPU experiment.  Whole
    runs, DoE and IOHprofiler files included (a temporary folder); barrier before and after, the job's time is the maximum over
    ranks, the aggregate the ranks' BO iterations over that time.  No data-path collective; the per-rank lines are gathered."""
    import tempfile
    from Algorithms import ExperimentRunner
    with tempfile.TemporaryDirectory(prefix="pcabo_bench_") as tmp:
        er = ExperimentRunner(algorithms=["pca"], dimensions=[20, 40], problem_ids=list(range(15, 25)), num_runs=30, root_dir=tmp,
                              experiment_name="bench", progress=False, batched=75, side_by_side=4, batch_acq_kernel="auto")
        mine = er._my_runs()
        torch.cuda.synchronize()
        D.barrier()
        t0 = time.perf_counter()
        er.run_experiment()
        torch.cuda.synchronize()
        dt_local = time.perf_counter() - t0
        D.barrier()
    iters = float(sum(r["iterations"] for r in er.results))
    dt = D.max_over_ranks(dt_local)
    total = D.sum_over_ranks(iters)
    cores = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 0)
    per_rank = D.gather_best([iters / dt_local, dt_local, float(len(mine)), cores, float(torch.get_num_threads()),
                              float(len(er.failed_runs))])
    return {"workload": "configs[3]: 30 runs x f15-f24 x d in {20,40} = 600 runs, this rank's share through ExperimentRunner("
                        "batched=75, side_by_side=4, batch_acq_kernel='auto')",
            "arithmetic_mode": {str(k): v for k, v in er.arithmetic_modes.items()},
            "seconds": dt, "aggregate_bo_iterations_per_s": total / dt, "bo_iterations": total,
            "per_rank": [{"bo_iterations_per_s": p[0], "seconds": p[1], "runs": int(p[2]), "cores_in_affinity_mask": int(p[3]),
                          "torch_threads": int(p[4]), "failed_runs": int(p[5])} for p in per_rank],
            "host_threads_per_rank": "one Python thread interleaves up to eight device-mode batches (four per dimension); 8 pool threads draw the next "
                                     "iteration's noise blocks and Sobol engines; no gang workers in device mode",
            "backend": D.backend_name(), "rccl_ranks": D.ranks_seen(),
            "note": "whole runs incl. DoE and IOHprofiler files; no data-path collective; the per-rank figures above were "
                    "exchanged with one all-gather after the timed region"}


def device_lbfgsb_roofline(block: dict, runs_per_batch: int) -> dict:
    """k_lbfgsb_group against the HBM peak, two ways, both stated: (a) from the KERNEL's time - the committed rocprofv3 kernel
    statistics of a 30-run device-mode batch (profiles/<round>/batch30_device_kernel_stats.csv) with the algorithmic bytes that
    same run reported (batch30_device_under_rocprof.json): `achieved` / `frac`; (b) over the wall time of this run's 4-batch block
    (all phases of the iterations included): `block_*`.  Algorithmic bytes per evaluation of a restart group: 8 n (n + 1) + 16 n k
    + 8 n (the triangles of R and of its transpose, the normalised points twice, alpha).  The bytes are served by the
    L2 / Infinity Cache, not by HBM: FETCH / WRITE per launch from the separate --pmc passes are quoted beside them."""
    import csv
    out = {"kernel": "k_lbfgsb_group", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "block_runs_in_flight": 4 * runs_per_batch, "block_achieved": block["lbfgsb_algorithmic_bytes"] / block["seconds"] / 1e9,
           "block_frac": block["lbfgsb_algorithmic_bytes"] / block["seconds"] / 1e9 / HBM_PEAK_GBS,
           "block_group_evaluations": block["lbfgsb_group_evaluations"], "block_algorithmic_bytes": block["lbfgsb_algorithmic_bytes"],
           "cache_level": "L2 / Infinity Cache resident (R, RT, ZnT of a run: <= 3.5 MB); HBM sees what the PMC passes count"}
    for rel in ("r04", "r03"):
        stats = os.path.join(ROOT, "profiles", rel, "batch30_device_kernel_stats.csv")
        run = os.path.join(ROOT, "profiles", rel, "batch30_device_under_rocprof.json")
        if not (os.path.exists(stats) and os.path.exists(run)):
            continue
        try:
            row = next(r for r in csv.DictReader(open(stats)) if r["Name"].startswith("k_lbfgsb_group"))
            meta = json.loads(open(run).read().strip().splitlines()[-1])
            sec = float(row["TotalDurationNs"]) * 1e-9
            out.update({"achieved": meta["lbfgsb_algorithmic_bytes"] / sec / 1e9, "frac": meta["lbfgsb_algorithmic_bytes"] / sec / 1e9 / HBM_PEAK_GBS,
                        "kernel_launches": int(row["Calls"]), "kernel_avg_ms": float(row["AverageNs"]) * 1e-6,
                        "kernel_share_of_gpu_time_percent": float(row["Percentage"]),
                        "group_evaluations": meta["lbfgsb_group_evaluations"], "algorithmic_bytes": meta["lbfgsb_algorithmic_bytes"],
                        "source": f"profiles/{rel}/batch30_device_kernel_stats.csv (rocprofv3 --kernel-trace --stats over tools/gpu_batch_clock.py 30 40 15 1 0 device)"})
            pmc = os.path.join(ROOT, "profiles", rel, "pmc_device_lbfgsb.json")
            if os.path.exists(pmc):
                kk = json.load(open(pmc))["kernels"].get("k_lbfgsb_group")
                if kk and kk["dispatches"]:
                    out["traffic_per_launch"] = {"fetch_bytes": kk["fetch_bytes"] / kk["dispatches"], "write_bytes": kk["write_bytes"] / kk["dispatches"],
                                                 "source": f"profiles/{rel}/pmc_device_lbfgsb.json (8-run batch; FETCH_SIZE x 2 per the gfx950 note, WRITE_SIZE)"}
        except Exception as e:     # noqa: BLE001
            out["kernel_time_error"] = str(e)
        break
    return out


def nearest_pmc(n_mean: float):
    """PMC traffic of the acquisition kernel at the shape of the committed table that is closest to the mean n of the
    timed steps (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied)."""
    for rel in ("r04", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rel, "pmc_traffic.json")
        try:
            pmc = json.load(open(path))
        except Exception:   # noqa: BLE001
            continue
        rows = []
        for key, v in pmc.items():
            if isinstance(v, dict) and "traffic_bytes" in v and key.startswith("n"):
                n = int(key[1:].split("_")[0])
                k = int(key.split("_")[1][1:])
                q = int(key.split("_")[2][1:])
                alg = 4.0 * n * (n + 1.0) + 8.0 * n * k + 8.0 * n + 8.0 * q * k + 8.0 * q * (1 + k)
                rows.append({"n": n, "k": k, "q": q, "traffic_bytes": v["traffic_bytes"], "algorithmic_bytes": alg,
                             "ratio": v["traffic_bytes"] / alg})
        if rows:
            best = min(rows, key=lambda r: abs(r["n"] - n_mean))
            return best, rows, f"profiles/{rel}/pmc_traffic.json"
    return None, [], None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=ITERS)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PCABO_BENCH_BATCH", "30")),
                    help="also measure B runs of the same cell advancing together (batched contexts; configs[2]: 30 runs "
                         "on one GPU; ~12 s); 0 = skip")
    ap.add_argument("--no-kchol-grid", action="store_true", help="skip the K(X,X)+Cholesky micro-benchmark grid (SURVEY 8d)")
    args = ap.parse_args()

    rank, local_rank, size = D.init()
    if size != args.gpus and size > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    device = local_rank if size > 1 else 0
    if os.environ.get("PCABO_BENCH_DEVICE") is not None:      # rehearsal override: all ranks on one device (gloo)
        device = int(os.environ["PCABO_BENCH_DEVICE"])
    torch.cuda.set_device(device)
    torch.set_num_threads(4)

    # ---- the job: one run of configs[1] per GPU, handed out by the product's own partitioner ---------------------
    runs = sharding.enumerate_runs([FID], [DIM], size)
    my_runs = sharding.assign_runs(runs, size)[rank]

    # ---- warmup: W iterations of a throw-away run (another instance) ---------------------------------------------
    if args.warmup > 0:
        w = Run(device, (FID, DIM, 29))
        for _ in range(args.warmup):
            w.step()
        w.close()

    # ---- timed region: exactly K BO iterations per rank, spread over the rank's run(s) ---------------------------
    timed_at = spread(min(args.steps, ITERS))
    at = set(timed_at)
    todo = args.steps
    elapsed_local, n_seen, states, closed = 0.0, [], [], []
    sample_at = {timed_at[int(i)] for i in np.linspace(0, len(timed_at) - 1, min(12, len(timed_at)))}   # CPU-baseline states
    run_idx = 0
    D.barrier()
    torch.cuda.synchronize()
    t_job = time.perf_counter()
    while todo > 0:
        if run_idx < len(my_runs):
            run = my_runs[run_idx]
        else:                                                   # K > 330: further instances, disjoint between ranks
            run = (FID, DIM, my_runs[0][2] + size * run_idx)
        r = Run(device, run)
        for it in range(ITERS):
            if it in at and todo > 0:
                if rank == 0 and run_idx == 0 and it in sample_at and not args.no_cpu_baseline:
                    states.append((np.vstack(r.opt.x_evals), np.array(r.opt.f_evals)))     # untimed host copies
                n_seen.append(r.n)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r.step()
                torch.cuda.synchronize()
                elapsed_local += time.perf_counter() - t0
                todo -= 1
                if todo == 0:
                    break
            else:
                r.step()
        closed.append(r.close())
        run_idx += 1
    torch.cuda.synchronize()
    D.barrier()
    job_wall = time.perf_counter() - t_job
    elapsed = D.max_over_ranks(elapsed_local)
    total_steps = D.sum_over_ranks(args.steps)
    gathered = D.gather_best([closed[0]["best"]])    # the one collective of the design (RCCL when size > 1)

    timing = {}
    for c in closed:
        for key, val in list(c["total_times"].items()) + [("optimize_acqf/" + k, v) for k, v in c["phase"].items()]:
            timing[key] = timing.get(key, 0.0) + val
        timing["lbfgsb_rounds"] = timing.get("lbfgsb_rounds", 0) + c["rounds"]

    # ---- roofline: second, profiled pass over the SAME iterations (HIP events on the context's stream) -----------
    roof, kchol, extra = None, None, {}
    if rank == 0 and not args.no_roofline:
        prof, ns, calib = profiled_pass(device, timed_at)
        n_mean = float(np.mean(ns))
        a = prof["acq_partial"]
        if a["launches"]:
            # THREE clocks on this kernel, all stated: HIP events as recorded (an event pair around the launch; includes the
            # command processor's own work between the two records), the same minus the reading of two events recorded back
            # to back (a lower bound: with a kernel in between part of that reading overlaps), and rocprofv3's dispatch
            # begin/end average from the committed summary of this command.  `achieved` / `frac` use the RAW event time -
            # the longest of the three, so the fraction is not flattered by the calibration.
            dur_cal = a["ms"] * 1e-3 / a["launches"]
            dur = dur_cal + calib["pair_ms"] * 1e-3
            byt = a["bytes"] / a["launches"]
            near, table, src = nearest_pmc(n_mean)
            rp = rocprof_summary()
            rounds = max(1, int(timing.get("lbfgsb_rounds", 0)))
            roof = {"kernel": "k_acq_fast<SLAB,NB,false> (acquisition value+gradient, one launch per L-BFGS-B round; the timed "
                              "region runs the same arithmetic as the RESIDENT instantiation <SLAB,NB,true>, see `timed_region_kernel`)",
                    "bound": "hbm", "achieved": byt / dur / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": byt / dur / 1e9 / HBM_PEAK_GBS,
                    "clocks_us": {"hip_events_raw": dur * 1e6, "hip_events_minus_back_to_back_pair": dur_cal * 1e6,
                                  "back_to_back_pair": calib["pair_ms"] * 1e3, "pair_around_empty_kernel": calib["empty_kernel_ms"] * 1e3,
                                  "rocprofv3_committed": rp["plain_avg_us"] if rp else None,
                                  "rocprofv3_source": rp["source"] if rp else None},
                    "frac_from_rocprofv3": (byt / (rp["plain_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp and rp["plain_avg_us"] else None,
                    "timed_region_kernel": {
                        "kernel": "k_acq_fast<SLAB,NB,true>: ONE resident launch per optimize call, evaluations fed through the mailbox",
                        "lbfgsb_rounds_in_job": rounds,
                        "us_per_round_host_clock": 1e6 * timing.get("optimize_acqf/lbfgsb", 0.0) / rounds,
                        "note": "host clock of the L-BFGS-B phase of ALL BO iterations of the job's runs / their rounds (host "
                                "step + mailbox round trip + evaluation); rocprofv3 sees the resident kernel as one dispatch "
                                "per call",
                        "rocprofv3_resident_avg_us_per_call": rp["resident_avg_us_per_call"] if rp else None},
                    "traffic": near["traffic_bytes"] if near else None,
                    "traffic_shape": {k: near[k] for k in ("n", "k", "q", "algorithmic_bytes", "ratio")} if near else None,
                    "traffic_table": table, "traffic_source": src,
                    "avg_launch_us": dur * 1e6, "launches": a["launches"], "algorithmic_bytes_per_launch": byt,
                    "achieved_tflops": a["flops"] / (a["ms"] * 1e-3) / 1e12,
                    "n_mean": n_mean,
                    "note": f"profiled pass over the same {len(ns)} BO iterations as the timed region (n = {ns[0]}..{ns[-1]}, "
                            f"mean {n_mean:.0f}); HIP events on the context's stream around every launch; `traffic` is the PMC "
                            "figure at the committed shape closest to that mean n, next to that shape's algorithmic bytes; "
                            "latency-bound kernel, R and ZnT stay L2 / Infinity-Cache resident"}
        for name in ("wpca", "gram", "cholesky", "root_inverse_alpha", "acq_large_batches"):
            g = prof[name]
            if g["launches"]:
                extra[name] = {"avg_us": g["ms"] * 1e3 / g["launches"], "calls": g["launches"],
                               "GBs": g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] else None,
                               "TFLOPs": g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] else None}
        g, c = prof["gram"], prof["cholesky"]
        if g["launches"] and c["launches"] and (g["ms"] + c["ms"]) > 0:
            sec = (g["ms"] + c["ms"]) * 1e-3
            tf = (g["flops"] + c["flops"]) / sec / 1e12
            kchol = {"kernels": "k_zstats + k_znorm + k_gram; k_chol_step, one launch per 64-wide panel (final update + panel + look-ahead)",
                     "bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / FP64_PEAK_TFLOPS, "hbm_GBs": (g["bytes"] + c["bytes"]) / sec / 1e9,
                     "hbm_frac": (g["bytes"] + c["bytes"]) / sec / 1e9 / HBM_PEAK_GBS,
                     "avg_us_per_iteration": sec * 1e6 / g["launches"], "iterations": g["launches"], "batch": 1,
                     "note": "single run: 47 MFLOP / 5 MB per iteration at n=450 - latency-bound (a dependency chain of "
                             "n pivots); the batched path (--batch) puts many runs' factorisations side by side"}

    # ---- beyond the headline: B runs of the same cell in lock-step, and K(X,X)+Cholesky on SURVEY 8(d)'s grid -------
    batch, grid = None, None
    if rank == 0 and size == 1 and args.batch > 1:
        from pcabo import batchrun
        batch = batchrun.bench_block(device, args.batch, FID, DIM)
        batch.pop("best_f", None)
        if args.batch >= 4:
            # the same runs as two lock-step batches side by side (one host thread each): one batch's host-paced L-BFGS-B
            # rounds overlap the other's launches and bookkeeping (ExperimentRunner(batched=, side_by_side=2))
            two = batchrun.bench_block(device, args.batch, FID, DIM, sub_batches=2)
            batch["side_by_side"] = {k: two[k] for k in ("sub_batches", "aggregate_bo_iterations_per_s", "seconds", "bo_iterations",
                                                          "host_phase_seconds", "retries", "failed_runs")}
            # 4 x B runs in flight (configs[2] / [3] hold 270 / 600 runs): host-paced (four batches, a host thread and two gang
            # workers each) against the device-resident optimiser (every restart group's L-BFGS-B inside one
            # launch, csrc/kernels_lbfgsb.hip; four batches interleaved on ONE host thread, no worker threads)
            keys = ("runs", "sub_batches", "aggregate_bo_iterations_per_s", "seconds", "bo_iterations", "host_phase_seconds",
                    "retries", "failed_runs")
            big = batchrun.bench_block(device, 4 * args.batch, FID, DIM, sub_batches=4)
            dev = batchrun.bench_block(device, 4 * args.batch, FID, DIM, sub_batches=4, acq_kernel="device", schedule="interleaved")
            dev2 = batchrun.bench_block(device, 8 * args.batch, FID, DIM, sub_batches=4, acq_kernel="device", schedule="interleaved")
            # the device-resident optimiser at 30 and 60 runs as well (one batch / two batches of B runs on one host thread)
            dev30 = batchrun.bench_block(device, args.batch, FID, DIM, acq_kernel="device", schedule="interleaved")
            dev60 = batchrun.bench_block(device, 2 * args.batch, FID, DIM, sub_batches=2, acq_kernel="device", schedule="interleaved")
            batch["one_batch_device_resident"] = {k: dev30[k] for k in keys}
            batch["two_batches_device_resident"] = {k: dev60[k] for k in keys}
            batch["four_batches_host_paced"] = {k: big[k] for k in keys}
            batch["four_batches_device_resident"] = {**{k: dev[k] for k in keys}, "host_thread_busy_seconds": dev["interleave"]["host_busy_seconds"],
                                                     "note": "pcabo.batchrun.run_interleaved + acq_kernel='device' (PCABO_OPT_DEVICE_LBFGSB = 1); a run "
                                                             "is bit-identical to the same run with the host's L-BFGS-B over the same evaluation "
                                                             "kernel (tests/test_gpu_device_lbfgsb.py)"}
            batch["four_double_batches_device_resident"] = {**{k: dev2[k] for k in keys},
                                                            "host_thread_busy_seconds": dev2["interleave"]["host_busy_seconds"]}
            # what the device-resident optimiser's evaluations move, algorithmically, against the chip's HBM peak: evaluations the
            # optimisers report x bytes of one evaluation at that iteration's (n, k), over the wall time of the whole block (all
            # phases of the iterations, not the kernel alone - an aggregate, beside `roofline` of the single run's kernel)
            batch["roofline_device_lbfgsb"] = device_lbfgsb_roofline(dev2, 2 * args.batch)
    if rank == 0 and size == 1 and not args.no_kchol_grid and not args.no_roofline:
        from pcabo import kchol_bench
        grid = kchol_bench.run(device, (1, 30), reps=3, big_batch=120)
        best = max((g for g in grid if (g["n"], g["k"]) == (450, 36) and g["batch"] <= 30), key=lambda g: g["kchol_tflops"])
        if kchol is not None:
            kchol["batched"] = {"n": best["n"], "k": best["k"], "batch": best["batch"], "achieved": best["kchol_tflops"],
                                "frac": best["kchol_frac_of_fp64_peak"], "hbm_GBs": best["kchol_GBs"],
                                "hbm_frac": best["kchol_frac_of_hbm_peak"], "us": best["us"],
                                "note": "same kernels, blockIdx.z = run: the headline shape with 30 runs' factorisations side by "
                                        "side (pcabo_batch_*); full grid in `kchol_grid` (batch 1 / 30, and 120 for the two larger shapes)"}
            big = [g for g in grid if g["batch"] == 120]
            if big:
                kchol["batched_120"] = [{"n": g["n"], "k": g["k"], "achieved": g["kchol_tflops"], "frac": g["kchol_frac_of_fp64_peak"],
                                         "us": g["us"]} for g in big]

    multi = None
    if size > 1 and args.batch > 1:
        multi = batch_per_gpu(device, rank, size, args.batch)

    cpu = None
    if rank == 0 and size == 1 and not args.no_cpu_baseline and states:      # N = 1 only (contract)
        # the oracle's tiny fp64 tensors run fastest single-threaded on this host (measured 0.246 s/iteration at 1
        # thread vs 0.465 s at 16, tools/cpu_oracle_threads.py), so the baseline gets its best setting
        cpu = cpu_baseline(states, threads=1)

    if rank == 0:
        value = total_steps / elapsed
        line = {
            "metric": "BO iterations/sec (incl. GP refit + EI opt), BBOB f15 d=40",
            "value": value, "unit": "BO iterations/s", "n_gpus": size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: PCA_BO on BBOB f15 d=40, budget=450, n_DoE=120, EI, var_threshold=0.95; "
                                   "one run per GPU (run list f15/d40/instances 0..N-1 partitioned by pcabo.sharding), "
                                   "timed steps spread evenly over the 330 BO iterations of the run",
                       "num_restarts": 10, "raw_samples": 512, "batch_limit": 5, "maxiter": 200,
                       "parallelism": f"run-parallel x{size}"},
            "n_range": [int(min(n_seen)), int(max(n_seen))], "n_mean": float(np.mean(n_seen)),
            "timed_iterations": len(n_seen), "timed_seconds": elapsed, "job_wall_seconds": job_wall,
            "backend": D.backend_name(),
            "roofline": roof, "roofline_kchol": kchol, "cpu_baseline": cpu, "batched": batch, "batched_per_gpu": multi,
            "rccl_ranks": D.ranks_seen(), "kchol_grid": grid,
            "kernels": extra, "host_phase_seconds": timing, "best_f": gathered,
            "runs": [list(r) for r in runs],
            "speedup_vs_cpu_baseline": (value / size / cpu["value"]) if cpu else None,
        }
        print(json.dumps(line), flush=True)
    D.barrier()          # ranks leave together (rank 0 ran the profiled pass meanwhile)
    D.finalize()


if __name__ == "__main__":
    main()
