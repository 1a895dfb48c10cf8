#!/usr/bin/env python3
"""Mean BFS / SSSP enact() and advance-kernel time over source 0 and five seeded random sources of
R-MAT-22 (whatever the environment's knobs select).  usage: source_mean.py [scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
deg = np.diff(g.offsets_to_host())
rng = np.random.default_rng(5)
sources = [0] + [int(x) for x in rng.choice(np.flatnonzero(deg > 0), 5, replace=False)]
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
o = ea.Options(collect_kernel_time=True)
be, bk, se, sk = [], [], [], []
for s in sources:
    for _ in range(3):
        _, st = ea.bfs(ctx, g, s, d, o)
    be.append(st.elapsed_ms); bk.append(st.advance_kernel_ms)
    for _ in range(3):
        _, st = ea.sssp(ctx, g, s, w, o)
    se.append(st.elapsed_ms); sk.append(st.advance_kernel_ms)
print(f"mean of {len(sources)} sources: BFS enact {np.mean(be):.3f} ms kernels {np.mean(bk):.3f} | SSSP enact {np.mean(se):.3f} ms "
      f"kernels {np.mean(sk):.3f} | BFS+SSSP enact {np.mean(be) + np.mean(se):.3f} ms")
