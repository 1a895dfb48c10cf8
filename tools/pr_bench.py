#!/usr/bin/env python3
"""PageRank (BASELINE config 4 stand-in): R-MAT scale-24 edgefactor-16 DIRECTED, alpha .85, tol 1e-6,
graph -> none advance.  Reports ms / iteration and edges/s per schedule, plus the CPU restatement."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--lbs", default="block_mapped,merge_path,bucketing")
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, a.scale, 16, 1, 0, symmetrize=False)
print(f"graph V={g.n_rows} E={g.nnz} (directed)", flush=True)
for lb in a.lbs.split(","):
    best = None
    for r in range(3):
        p, st = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(load_balance=ea.LoadBalance[lb], collect_kernel_time=True))
        if best is None or st.elapsed_ms < best.elapsed_ms: best = st
    it = best.iterations
    nbytes = (12 * g.nnz + 44 * g.n_rows) * it
    print(f"pr {lb:13s} {best.elapsed_ms:8.2f} ms {it} iters  {best.elapsed_ms/it:7.3f} ms/iter  advance {best.advance_kernel_ms/it:7.3f} ms/iter "
          f"{g.nnz*it/best.elapsed_ms/1e6:7.2f} GTEPS  algorithmic {nbytes/best.elapsed_ms/1e6:7.1f} GB/s  sum={float(p.sum()):.6f}", flush=True)
g.build_in_edges(ctx)   # transpose for the pull form
best = None
for r in range(3):
    p, st = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
    if best is None or st.elapsed_ms < best.elapsed_ms: best = st
it = best.iterations
nbytes = (12 * g.nnz + 44 * g.n_rows) * it
print(f"pr PULL          {best.elapsed_ms:8.2f} ms {it} iters  {best.elapsed_ms/it:7.3f} ms/iter  "
      f"{g.nnz*it/best.elapsed_ms/1e6:7.2f} GTEPS  algorithmic {nbytes/best.elapsed_ms/1e6:7.1f} GB/s  sum={float(p.sum()):.6f}", flush=True)
if a.cpu:
    from oracle.oracle import Oracle
    o = Oracle(); Ap, Aj, Ax = g.to_host()
    t0 = time.time(); want, it = o.pagerank(Ap, np.ascontiguousarray(Aj), np.ascontiguousarray(Ax), 0.85, 1e-6); dt = time.time() - t0
    print(f"cpu restatement: {dt*1e3:.0f} ms {it} iters {g.nnz*it/dt/1e6:.1f} MTEPS  max|gpu-cpu|={np.abs(p.cpu().numpy()-want).max():.2e}")
