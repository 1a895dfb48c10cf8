#!/usr/bin/env python3
"""Direction-optimising BFS vs push-only BFS on R-MAT: time, levels pulled, sweep of alpha/beta."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--alphas", default="14")
ap.add_argument("--betas", default="24")
ap.add_argument("--sources", default="0,2836928,911684,216248")
a = ap.parse_args()
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, a.scale, 16, 1, 7)
srcs = [int(x) % g.n_rows for x in a.sources.split(",")]
def best(src, **kw):
    b = None
    for _ in range(4):
        d, st = ea.bfs(ctx, g, src, options=ea.Options(collect_kernel_time=False, **kw))
        if b is None or st.elapsed_ms < b.elapsed_ms: b = st
    return b
for s in srcs:
    p = best(s)
    print(f"src {s:8d} push   {p.elapsed_ms:6.3f} ms {p.edges_traversed/p.elapsed_ms/1e6:7.1f} GTEPS levels {p.frontier_slots}", flush=True)
    for al in [float(x) for x in a.alphas.split(",")]:
        for be in [float(x) for x in a.betas.split(",")]:
            q = best(s, direction_optimized=True, do_alpha=al, do_beta=be)
            print(f"             do a={al:5.1f} b={be:5.1f} {q.elapsed_ms:6.3f} ms {q.edges_traversed/q.elapsed_ms/1e6:7.1f} GTEPS pulls {q.pull_iterations} levels {q.frontier_slots}", flush=True)
