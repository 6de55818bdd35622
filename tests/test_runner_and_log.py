"""ExperimentRunner mirror and the IOHprofiler-format writer: host logic only, no GPU."""
import json
import os

import numpy as np

from pcabo import iohlog, sharding
from pcabo.bbob import BBOBProblem

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats_dim5.json")))


def _log_one(tmp_path, fid=15, inst=0, n=12):
    lg = iohlog.Analyzer(root=str(tmp_path), folder_name="pca-experiment", algorithm_name="pca",
                         algorithm_info="A pca-BO Implementation.")
    lg.set_experiment_attributes({"budget_factor": "5", "doe_factor": "2.0", "acquisition_function": "expected_improvement"})
    lg.set_experiment_attributes({"pca_components": "0", "var_threshold": "0.95"})
    for name in ("SingleTaskGP", "optimize_acqf", "pca"):
        lg.add_run_attribute(f"{name}_time", 0.0)
    lg.add_run_attribute("time", 0.0)
    p = iohlog.LoggedProblem(BBOBProblem(fid, inst, 5), lg)
    doe = [d for d in G["doe"] if d["fid"] == fid and d["instance"] == inst and d["alg"] == "pca"][0]
    ys = [p(np.array(x)) for x in doe["x"]]
    rng = np.random.default_rng(0)
    ys += [p(rng.uniform(-5, 5, 5)) for _ in range(n - len(ys))]
    lg.set_run_attribute("time", 1.25)
    lg.set_run_attribute("pca_time", 0.5)
    return lg, p, ys


def test_dat_layout_matches_reference_files(tmp_path):
    lg, p, ys = _log_one(tmp_path)
    lg.close()
    path = os.path.join(lg.output_directory, "data_f15_RastriginRotated", "IOHprofiler_f15_DIM5.dat")
    lines = open(path).readlines()
    assert lines[0] == G["dat_header"]                       # identical column header
    assert lines[1] == G["dat_first_row"]                    # identical first row: same x, raw_y, formatting
    rows = iohlog.read_dat(path)[0]
    assert rows.shape == (12, 8) and (rows[:, 0] == np.arange(1, 13)).all()
    assert np.allclose(rows[:, 2], np.minimum.accumulate(rows[:, 1]))                 # raw_y_best column
    assert np.allclose(rows[:, 1] + p._f_opt, ys, atol=1e-9)                         # raw_y has no f_opt shift


def test_json_layout_matches_reference_files(tmp_path):
    lg, p, ys = _log_one(tmp_path)
    lg.close()
    meta = json.load(open(os.path.join(lg.output_directory, "IOHprofiler_f15_RastriginRotated.json")))
    assert list(meta.keys()) == G["json_keys"]
    assert {k: meta[k] for k in meta if k != "scenarios"} == G["json_head"]           # incl. replaced experiment attrs
    sc = meta["scenarios"][0]
    assert list(sc.keys()) == G["json_scenario_keys"] and sc["dimension"] == 5
    assert sc["path"] == "data_f15_RastriginRotated/IOHprofiler_f15_DIM5.dat"
    run = sc["runs"][0]
    assert list(run.keys()) == G["json_run_keys"]
    assert run["evals"] == 12 and run["time"] == 1.25 and run["pca_time"] == 0.5 and run["SingleTaskGP_time"] == 0
    raw = np.array(ys) - p._f_opt
    assert run["best"]["evals"] == int(np.argmin(raw)) + 1 and abs(run["best"]["y"] - raw.min()) < 1e-9
    assert len(run["best"]["x"]) == 5


def test_logger_never_overwrites_and_appends_runs(tmp_path):
    lg, _, _ = _log_one(tmp_path)
    iohlog.LoggedProblem(BBOBProblem(15, 1, 5), lg)(np.zeros(5))
    iohlog.LoggedProblem(BBOBProblem(20, 0, 5), lg)(np.zeros(5))
    lg.close()
    lg2, _, _ = _log_one(tmp_path)
    lg2.close()
    assert lg2.output_directory.endswith("pca-experiment-1")
    meta = json.load(open(os.path.join(lg.output_directory, "IOHprofiler_f15_RastriginRotated.json")))
    assert [r["instance"] for r in meta["scenarios"][0]["runs"]] == [0, 1]
    assert os.path.exists(os.path.join(lg.output_directory, "IOHprofiler_f20_Schwefel.json"))
    blocks = iohlog.read_dat(os.path.join(lg.output_directory, "data_f15_RastriginRotated", "IOHprofiler_f15_DIM5.dat"))
    assert [len(b) for b in blocks] == [12, 1]


def test_runner_shards_cover_all_runs_once(monkeypatch):
    """Every (function, dimension, instance) lands on exactly one rank; single process keeps the suite order."""
    from Algorithms import ExperimentRunner
    kw = dict(algorithms=["pca", "vanilla"], dimensions=[5, 10], problem_ids=[15, 20], num_runs=7,
              budget_factor=5, doe_factor=2.0, progress=False)
    monkeypatch.delenv("RANK", raising=False); monkeypatch.delenv("WORLD_SIZE", raising=False)
    single = ExperimentRunner(**kw)
    assert single._my_runs() == sharding.enumerate_runs([15, 20], [5, 10], 7) and single._folder("pca") == "pca-experiment"
    seen = []
    for r in range(3):
        monkeypatch.setenv("RANK", str(r)); monkeypatch.setenv("WORLD_SIZE", "3"); monkeypatch.setenv("LOCAL_RANK", str(r))
        er = ExperimentRunner(**kw)
        assert er.device == r and er._folder("vanilla") == f"vanilla-experiment-rank{r}"
        seen += er._my_runs()
    assert sorted(seen) == sorted(single._my_runs()) and len(set(seen)) == len(seen)
    try:
        ExperimentRunner(**{**kw, "algorithms": ["cma"]}).run_experiment()
        raise AssertionError("invalid algorithm accepted")
    except ValueError as e:
        assert "Invalid algorithm name" in str(e)
