"""pr.hxx's push PageRank (whole-graph advance without an output) row by row and grouped by
destination (operators/by_destination.hxx): ms per iteration on a directed R-MAT, ranks against
the pull formulation.  Usage: python3 tools/pr_by_destination.py [scale]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import essentials_amd as ea

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 0, False)
print(f"directed R-MAT scale {scale}: V={g.n_rows} E={g.nnz}", flush=True)
runs = []
for r in range(4):
    t0 = time.perf_counter()
    p, st = ea.pagerank(ctx, g, 0.85, 1e-6)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    runs.append((st.elapsed_ms, st.iterations, wall))
    print(f"run {r}: enact {st.elapsed_ms:.2f} ms, {st.iterations} iterations = "
          f"{st.elapsed_ms / st.iterations:.2f} ms/iteration (wall {wall:.1f} ms), sum {float(p.double().sum()):.6f}",
          flush=True)
for stream in ("0", "1", "0", "1"):
    os.environ["GRX_BY_DESTINATION_STREAM"] = stream
    p, st = ea.pagerank(ctx, g, 0.85, 1e-6)
    print(f"push, list loaded {'streaming' if stream == '1' else 'plain'}: "
          f"{st.elapsed_ms / st.iterations:.2f} ms/iteration", flush=True)
g.build_in_edges(ctx)
for walk in ("1", "0"):
    os.environ["GRX_PR_PULL_WALK"] = walk
    for r in range(3):
        os.environ["GRX_BY_DESTINATION_STREAM"] = "01"[r % 2]
        q, sq = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
        print(f"pull ({'sorted-list walk' if walk == '1' else 'per-destination lists'}) run {r}: "
              f"{sq.elapsed_ms / sq.iterations:.2f} ms/iteration, {sq.iterations} iterations; "
              f"max |push - pull| = {float((p - q).abs().max()):.3e}", flush=True)
