#!/usr/bin/env python3
"""BFS advance-kernel time of the settled wide levels against hub threshold / chunk size / the work
from which the form is taken (GRX_SETTLED_MIN_WORK is read when the context is made: one process
per value).  usage: settled_tune.py [scale] [n_sources]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n_src = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
deg = np.diff(g.offsets_to_host())
rng = np.random.default_rng(5)
sources = [0] + rng.choice(np.flatnonzero(deg > 0), n_src - 1).tolist()
print("GRX_SETTLED_MIN_WORK", os.environ.get("GRX_SETTLED_MIN_WORK", "(default)"))
for hub, chunk in ((0, 0), (128, 1024), (512, 1024), (256, 512), (256, 2048), (128, 512), (512, 2048)):
    tot_k = tot_e = 0.0
    for s in sources:
        best = None
        for _ in range(4):
            _, st = ea.bfs(ctx, g, int(s), d, ea.Options(collect_kernel_time=True, hub_threshold=hub, chunk_edges=chunk))
            if best is None or st.advance_kernel_ms < best.advance_kernel_ms:
                best = st
        tot_k += best.advance_kernel_ms
        tot_e += best.elapsed_ms
    print(f"hub {hub or 256:4d} chunk {chunk or 1024:5d}: kernels {tot_k / len(sources):.3f} ms  enact {tot_e / len(sources):.3f} ms (mean over {len(sources)} sources)", flush=True)
