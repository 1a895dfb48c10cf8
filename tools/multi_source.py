#!/usr/bin/env python3
"""Mean enact time of BFS and SSSP over source 0 and 15 seeded random non-isolated sources on
RMAT-22 (what the bench's steps average over), best of 3 per source."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, 22, 16, 1, 7)
deg = np.diff(g.offsets_to_host())
rng = np.random.default_rng(100)
sources = [0] + [int(x) for x in rng.choice(np.flatnonzero(deg > 0), 15)]
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
for name, fn, buf in (("bfs", ea.bfs, d), ("sssp", ea.sssp, w)):
    tot, its = 0.0, 0
    for s in sources:
        best = None
        for _ in range(3):
            _, st = fn(ctx, g, s, buf)
            if best is None or st.elapsed_ms < best.elapsed_ms:
                best = st
        tot += best.elapsed_ms
        its += best.iterations
    print(f"{name:4s} mean enact {tot/len(sources):.3f} ms over {len(sources)} sources ({its/len(sources):.1f} iterations on average)", flush=True)
