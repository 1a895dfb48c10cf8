#!/usr/bin/env python3
"""PageRank soak: on random R-MAT graphs (scale, edge factor, seed, directed or not) the push form
walked by destination (forced from the second iteration on), the push form walked row by row and
the pull form must agree within 5e-6 and sum to 1.   usage: soak_pagerank.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, runs, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    scale = int(rng.integers(4, 19))
    ef = int(rng.choice([1, 2, 8, 16, 37]))
    sym = bool(rng.integers(0, 2))
    seed = int(rng.integers(1, 10000))
    ranks = {}
    # one weight seed for both modes
    ws = int(rng.integers(0, 9))
    lb = ea.LoadBalance[str(rng.choice(["block_mapped", "merge_path", "bucketing"]))]
    for mode in ("1", "0"):
        os.environ["GRX_BY_DESTINATION"] = mode
        ctx = ea.Context(0)
        g = ea.Graph.rmat(ctx, scale, ef, seed, ws, sym)
        for again in range(2):
            p, st = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(load_balance=lb))
        ranks[mode] = p.clone()
        if mode == "1":
            if not sym:
                g.build_in_edges(ctx)
            q, _ = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
            ranks["pull"] = q.clone()
        g.close()
        ctx.close()
    ok = abs(float(ranks["1"].double().sum()) - 1.0) < 1e-3
    ok &= float((ranks["1"] - ranks["0"]).abs().max()) < 5e-6
    ok &= float((ranks["1"] - ranks["pull"]).abs().max()) < 5e-6
    runs += 1
    if not ok:
        bad += 1
        print("MISMATCH", scale, ef, sym, seed, ws, lb.name, float((ranks["1"] - ranks["0"]).abs().max()),
              float((ranks["1"] - ranks["pull"]).abs().max()), flush=True)
print(f"pagerank soak: {runs} graphs, {bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
