#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy of every kernel in one HIP source (hipcc -Rpass-analysis)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only",
       "-I", os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"gunrock::|essentials_amd::|hip::kernels::|operators::", "", name)
    short = re.sub(r"graph::graph_t<[^>]*>", "G", short)
    print(f'{short[:150]:150s} vgpr={r.get("VGPRs")} sgpr={r.get("TotalSGPRs")} scratch={r.get("ScratchSize [bytes/lane]")} lds={r.get("LDS Size [bytes/block]")} occ={r.get("Occupancy [waves/SIMD]")}')
