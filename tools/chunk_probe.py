#!/usr/bin/env python3
"""Narrow levels that consist of a few hub lists (level 0 of a hub source: ONE vertex, 320 K edges):
kernel times of the first levels with 1024- and 256-edge chunks.  Run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import essentials_amd as ea
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, 22, 16, 1, 7)
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
ce = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(3):
    _, st = ea.bfs(ctx, g, 0, d, ea.Options(collect_kernel_time=True, chunk_edges=ce, max_iterations=2))
print("chunk_edges", ce, "enact ms", st.elapsed_ms, "kernels ms", st.advance_kernel_ms)
