"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel (and per grid for k_acq_fused)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (name, r["Grid_Size"], r["Counter_Name"])
        acc[key].append(float(r["Counter_Value"]))
    for (name, grid, ctr), v in sorted(acc.items()):
        if name.startswith(("k_acq", "k_read8", "k_write8", "k_gram", "k_chol", "k_trinv", "k_jacobi")):
            print(f"{name:22s} grid={grid:>8s} {ctr:12s} n={len(v):5d} mean={sum(v)/len(v):14.1f}")
