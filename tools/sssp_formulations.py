#!/usr/bin/env python3
"""One-pass (exact dedupe in the relax functor) against two-pass (advance + bypass filter, the
reference client's formulation) SSSP on RMAT-22 over several sources: enact ms, relaxations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, 22, 16, 1, 7)
deg = np.diff(g.offsets_to_host())
rng = np.random.default_rng(100)
sources = [0] + [int(x) for x in rng.choice(np.flatnonzero(deg > 0), 9)]
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
tot = {False: 0.0, True: 0.0}
for s in sources:
    row = []
    for two in (False, True):
        best = None
        for _ in range(3):
            _, st = ea.sssp(ctx, g, s, w, ea.Options(sssp_two_pass=two))
            if best is None or st.elapsed_ms < best.elapsed_ms:
                best = st
        tot[two] += best.elapsed_ms
        row.append(f"{'two' if two else 'one'}-pass {best.elapsed_ms:6.3f} ms {best.iterations:2d} it {best.edges_expanded/1e6:6.1f} M relax")
    print(f"source {s:8d}: " + " | ".join(row), flush=True)
print(f"mean one-pass {tot[False]/len(sources):.3f} ms, two-pass {tot[True]/len(sources):.3f} ms")
