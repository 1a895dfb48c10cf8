#!/usr/bin/env python3
"""How far is the advance skeleton from the random-gather roof?  One advance over ALL vertices of an
R-MAT graph with the BFS functor in three states of the label array:
  reject   labels = 0      every edge ends in the pre-test load (pure gather, no atomics, no output)
  discover labels = INT_MAX first arrival per vertex wins: V atomics + V outputs, E - V rejects
beside grx_measure_gather_rate (table[column[i]] over the same column array)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import essentials_amd as ea

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--lbs", default="block_mapped")
ap.add_argument("--hub", default="0")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, a.scale, 16, 1, 7)
V, E = g.n_rows, g.nnz
print(f"graph V={V} E={E}")
for mode in (1, 0):
    print(f"gather roof ({'agent-scope' if mode else 'plain'} loads): {ctx.gather_rate(g, mode)/1e9:6.1f} G lookups/s", flush=True)
dev = "cuda"
ordered = torch.arange(V, dtype=torch.int32, device=dev)
shuffled = ordered[torch.randperm(V, device=dev)].contiguous()
labels = torch.empty(V, dtype=torch.int32, device=dev)
for lb in a.lbs.split(","):
  for hub in [int(x) for x in a.hub.split(",")]:
    opts = ea.Options(load_balance=ea.LoadBalance[lb], hub_threshold=hub)
    for fname, frontier in (("ordered", ordered), ("shuffled", shuffled)):
        for state, fill in (("reject", 0), ("discover", ea.INT_UNREACHED)):
            best = 1e9
            for r in range(a.reps):
                labels.fill_(fill)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = ea.advance(ctx, g, frontier, ea.EdgeOp.bfs, labels, 5, opts, True, V + 1024)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print(f"{lb:14s} hub {hub:5d} {fname:9s} {state:9s} {best*1e3:7.3f} ms  {E/best/1e9:6.1f} G edges/s  out {out.numel()}", flush=True)
ap_host = g.to_host()[0]
import numpy as np
deg = np.diff(ap_host).astype(np.int64)
for thr in (64, 256, 1024):
    print(f"edges in rows of degree >= {thr}: {deg[deg >= thr].sum()/E*100:5.1f} %  ({(deg >= thr).sum()} rows); zero-degree rows {(deg == 0).sum()}")
