#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name prefix) and dispatch."""
import csv, glob, re, sys, collections
d = sys.argv[1]
want = sys.argv[2:] or ["block_mapped_kernel", "chunk_kernel", "gather_probe"]
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in rows:
        k = re.sub(r"<.*", "", r["Kernel_Name"]).split("::")[-1].replace("void ", "")
        if not any(w in k for w in want):
            continue
        by.setdefault((r["Dispatch_Id"], k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for (did, k), c in by.items():
        print(did, k[:24], " ".join(f"{n}={v:.4g}" for n, v in c.items()))
