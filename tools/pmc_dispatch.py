#!/usr/bin/env python3
"""Per-dispatch counter values of the kernels matching a name, from `rocprofv3 --pmc ... --output-format
csv -d DIR`: one line per dispatch, launch order.  usage: pmc_dispatch.py DIR kernel_substring [last_n]"""
import collections, csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
want = sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if want in r["Kernel_Name"]:
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(rows)[-last:]:
    print(d, " ".join(f"{k}={v:.4g}" for k, v in sorted(rows[d].items())))
