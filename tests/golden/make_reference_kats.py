"""Extract the known answers the reference ships into small committed fixtures.

Run in the build container (the only place /root/reference exists):
    python tests/golden/make_reference_kats.py
Outputs tests/golden/ref_kats_dim5.json holding DATA only (inputs + expected outputs):
  * "doe":  for each of the 120 committed runs (pca/vanilla x f15/f20 x 30 instances, d=5,
            n_DoE=10) the seed and the 10 DoE rows (x printed to 1e-6 by the IOH logger)
            -> pins seed formula (ExperimentRunner.py:146) + LHS "center" draw.
  * "f15_doe": raw_y of the f15 DoE rows (x exact: bin centres) -> pins BBOB f15 to ~1e-12.
  * "f15_best": full-precision best x / y per run from the two f15 .json files
            -> pins BBOB f15 on off-grid points to ~1e-13.
  * "f15_bo_rows": a sample of BO-phase rows (x printed to 1e-6) -> pins f15 to ~1e-5.
  * "f20_doe", "f20_best", "f20_bo_rows": the same three for BBOB f20 (Schwefel), the second function of the
            reference's --quick configuration (main.py:103-109).
  * "final_best": min raw_y of every committed run (alg, fid, instance) -> the end-to-end statistical check of
            Vanilla_BO (the PCA_BO files come from an older, clipping revision of the reference: SURVEY.md fact 6).
  * "vanilla_runs": all 75 logged rows (raw_y, x printed to 1e-6) of 12 committed Vanilla_BO runs -> each BO row is the
            candidate the REFERENCE chose given the rows before it: it must be a local maximum of the acquisition
            surface built from those rows (pins Standardize / Matern-5/2 / lengthscale / noise / log-EI / best_f).
  * ref_vanilla_runs_dim5.json (second file): ALL 60 committed Vanilla_BO runs (f15 and f20, instances 0..29), 75 rows each,
            same row format as "vanilla_runs" -> the local-optimum check over every BO row the reference logged.
  * "dat_header", "json_keys": the layout of the IOHprofiler 0.3.18 files (column header of a .dat block, key order
            of the .json) -> pins the writer in pcabo/iohlog.py.
Source files: /root/reference/{pca,vanilla}-experiment/data_f*/IOHprofiler_f*_DIM5.dat and
/root/reference/{pca,vanilla}-experiment/IOHprofiler_f15_RastriginRotated.json
"""
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats_dim5.json")


def read_runs(path):
    runs, cur = [], None
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "evaluations":
            cur = []
            runs.append(cur)
        else:
            cur.append([float(v) for v in t])
    return runs


def main():
    out = {"doe": [], "f15_doe": [], "f15_best": [], "f15_bo_rows": [], "f20_doe": [], "f20_best": [], "f20_bo_rows": []}
    all_vanilla = []          # every committed Vanilla_BO run in full -> ref_vanilla_runs_dim5.json (round 2)
    for alg in ("pca", "vanilla"):
        for fid, name in ((15, "RastriginRotated"), (20, "Schwefel")):
            meta = json.load(open(f"{REF}/{alg}-experiment/IOHprofiler_f{fid}_{name}.json"))
            insts = [r["instance"] for r in meta["scenarios"][0]["runs"]]
            runs = read_runs(f"{REF}/{alg}-experiment/data_f{fid}_{name}/IOHprofiler_f{fid}_DIM5.dat")
            # the .dat holds one block per run in instance order 0..29 (the .json may list fewer)
            for inst, run in enumerate(runs):
                first = run[0][0]
                doe = [r for r in run[:10]]
                out["doe"].append({"alg": alg, "fid": fid, "dim": 5, "instance": inst,
                                   "seed": 1000 * fid + 10 * 5 + inst, "first_eval_index": first,
                                   "x": [r[3:] for r in doe]})
                out.setdefault("final_best", []).append({"alg": alg, "fid": fid, "instance": inst, "rows": len(run),
                                                         "best": min(r[1] for r in run)})
                if alg == "vanilla":
                    all_vanilla.append({"fid": fid, "instance": inst, "rows": [[r[1]] + r[3:] for r in run]})
                if alg == "vanilla" and inst % 5 == 0:
                    out.setdefault("vanilla_runs", []).append({"fid": fid, "instance": inst,
                                                               "rows": [[r[1]] + r[3:] for r in run]})
                out[f"f{fid}_doe"].append({"alg": alg, "instance": inst, "raw_y": [r[1] for r in doe]})
                for r in run[10::13]:
                    out[f"f{fid}_bo_rows"].append({"instance": inst, "x": r[3:], "raw_y": r[1]})
            for r in meta["scenarios"][0]["runs"]:
                out[f"f{fid}_best"].append({"alg": alg, "instance": r["instance"],
                                            "x": r["best"]["x"], "y": r["best"]["y"]})
            if alg == "pca" and fid == 15:
                out["dat_header"] = open(f"{REF}/{alg}-experiment/data_f{fid}_{name}/IOHprofiler_f{fid}_DIM5.dat").readline()
                out["dat_first_row"] = open(f"{REF}/{alg}-experiment/data_f{fid}_{name}/IOHprofiler_f{fid}_DIM5.dat").readlines()[1]
                out["json_keys"] = list(meta.keys())
                out["json_scenario_keys"] = list(meta["scenarios"][0].keys())
                out["json_run_keys"] = list(meta["scenarios"][0]["runs"][0].keys())
                out["json_head"] = {k: meta[k] for k in meta if k != "scenarios"}
    json.dump(out, open(OUT, "w"))
    print(OUT, os.path.getsize(OUT), "bytes;", {k: len(v) for k, v in out.items()})
    out2 = os.path.join(os.path.dirname(OUT), "ref_vanilla_runs_dim5.json")
    json.dump({"vanilla_runs": all_vanilla}, open(out2, "w"))
    print(out2, os.path.getsize(out2), "bytes;", len(all_vanilla), "runs")


if __name__ == "__main__":
    main()
