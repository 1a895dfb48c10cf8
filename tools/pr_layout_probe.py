"""PageRank on the generated numbering of a directed R-MAT against the same graph renumbered by
falling out-degree (what the C ABI's hot-first copy is) and scrambled: ms per iteration of the push
form (two lookups by source per edge) and the pull form (one), both on the destination-sorted walk.
Usage: python3 tools/pr_layout_probe.py [scale]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import essentials_amd as ea
from layout_lib import relabelled

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = ea.Context(0)
g0 = ea.Graph.rmat(ctx, scale, 16, 1, 0, False)
for layout in ("generated", "degree-ordered", "scrambled"):
    g, _ = relabelled(ctx, g0, layout)
    for r in range(2):
        p, st = ea.pagerank(ctx, g, 0.85, 1e-6)
    g.build_in_edges(ctx)
    for r in range(2):
        q, sq = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
    print(f"{layout:15s} push {st.elapsed_ms / st.iterations:.2f} ms/iteration ({st.iterations}), "
          f"pull {sq.elapsed_ms / sq.iterations:.2f} ms/iteration ({sq.iterations})", flush=True)
    if g is not g0:
        g.close()
