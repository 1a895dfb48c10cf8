#!/usr/bin/env python3
"""One BFS (or SSSP) from source 0 on a relabelled copy of the R-MAT graph, three times: run under
`rocprofv3 --kernel-trace` and read the last traversal with tools/trace_levels.py.
usage: level_probe.py LAYOUT [bfs|sssp] [scale]      LAYOUT = generated | degree-ordered | scrambled"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import essentials_amd as ea
from layout_lib import relabelled
layout = sys.argv[1] if len(sys.argv) > 1 else "generated"
algo = sys.argv[2] if len(sys.argv) > 2 else "bfs"
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 22
ctx = ea.Context(0)
g0 = ea.Graph.rmat(ctx, scale, 16, 1, 7)
g, rank = relabelled(ctx, g0, layout)
src = 0 if rank is None else int(rank[0])
out = torch.empty(g.n_rows, dtype=torch.int32 if algo == "bfs" else torch.float32, device="cuda")
o = ea.Options(collect_kernel_time=True, call_every_edge=bool(int(os.environ.get("EVERY_EDGE", "0"))))
for _ in range(3):
    _, st = (ea.bfs if algo == "bfs" else ea.sssp)(ctx, g, src, out, o)
print(layout, algo, "enact ms", st.elapsed_ms, "kernels ms", st.advance_kernel_ms, "slots", st.frontier_slots[:10],
      "edges expanded", st.edges_expanded)
