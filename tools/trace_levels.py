#!/usr/bin/env python3
"""Kernel sequence of the LAST traversal in a `rocprofv3 --kernel-trace` of tools/level_profile.py:
name (template arguments cut) and duration of every dispatch, in launch order."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("<")[0].split("(")[0].split("::")[-1] for r in rows]
# last traversal = from the last reset pass of a problem (index_kernel over `...problem_t<...>::reset()`'s
# lambda) on; the default forms no longer end with a reach_stats_kernel (round 3)
starts = [i for i, r in enumerate(rows) if "problem_t<" in r["Kernel_Name"] and "::reset()" in r["Kernel_Name"]]
lo = starts[-1] if starts else 0
hi = len(rows)
tot = 0.0
for r, n in zip(rows[lo:hi], names[lo:hi]):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    print(f"{n:34s} {us:9.1f} us  grid {r.get('Grid_Size','?'):>8s} wg {r.get('Workgroup_Size','?')}")
print("sum", round(tot, 1), "us")
