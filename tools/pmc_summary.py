"""Collect rocprofv3 --pmc csv outputs (counter_collection.csv) under a directory into one table: per probe/kernel
the per-dispatch mean of every counter (our kernels only)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    probe = os.path.relpath(f, root).split(os.sep)[0].split("_")[0]
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "anonymous namespace" not in k:
            continue
        short = k.split("::")[1].split("(")[0][:60] if "::" in k else k[:60]
        acc[(probe, short)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (probe, k), cs in sorted(acc.items()):
    parts = [f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())]
    print(f"{probe:10s} {k:62s} " + " ".join(parts))
