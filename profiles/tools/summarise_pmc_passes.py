"""Summarise the rocprofv3 counter passes of tools/gpu_pmc_shapes.py.

  python profiles/tools/summarise_pmc_passes.py <dir with pmc_mfma/ pmc_fetch/ pmc_write/ subdirs> > profiles/r02/pmc_summary.txt

Per (kernel, grid size): mean duration from the kernel trace of the pmc_mfma pass; f64 MFMA flops = 512 *
SQ_INSTS_VALU_MFMA_MOPS_F64 (rocprofv3's MfmaFlopsF64 expression); MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE * 1024 SIMDs) (rocprofv3's MfmaUtil expression); fabric bytes = 2 * FETCH_SIZE * 1024 (gfx950: FETCH_SIZE
tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE * 1024, each from its own pass."""
import csv, glob, json, sys
from collections import defaultdict

root = sys.argv[1]
KEEP = ("k_gram", "k_chol", "k_trinv", "k_cov", "k_score", "k_acq", "k_jacobi", "k_rmatvec", "k_rtmatvec", "k_wpca", "k_project",
        "k_znorm", "k_zstats")


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(sub):
    acc = defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            acc[(short(r["Kernel_Name"]), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return acc


mf, fe, wr, du = counters("pmc_mfma"), counters("pmc_fetch"), counters("pmc_write"), durations("pmc_mfma")
mean = lambda v: sum(v) / len(v) if v else None
rows = []
for key in sorted(set(mf) | set(fe) | set(wr)):
    name, grid = key
    if not name.startswith(KEEP):
        continue
    m = mf.get(key, {})
    mops, busy, gui = mean(m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", [])), mean(m.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), mean(m.get("GRBM_GUI_ACTIVE", []))
    us = mean(du.get(key, []))
    fetch, write = mean(fe.get(key, {}).get("FETCH_SIZE", [])), mean(wr.get(key, {}).get("WRITE_SIZE", []))
    row = {"kernel": name, "grid_threads": grid, "calls": len(m.get("GRBM_GUI_ACTIVE", [])) or None, "us": us,
           "mfma_f64_flops": 512 * mops if mops is not None else None,
           "mfma_tflops": (512 * mops / (us * 1e-6) / 1e12) if mops and us else None,
           "mfma_util_pct": (100.0 * busy / (gui * 1024)) if busy is not None and gui else None,
           "fabric_bytes": (2 * fetch * 1024 + write * 1024) if fetch is not None and write is not None else None,
           "fetch_kb_raw": fetch, "write_kb": write}
    if row["fabric_bytes"] and us:
        row["fabric_GBs"] = row["fabric_bytes"] / (us * 1e-6) / 1e9
    rows.append(row)
print(json.dumps(rows, indent=1))
