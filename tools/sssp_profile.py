#!/usr/bin/env python3
"""Per-iteration work of one SSSP (frontier slots, edges relaxed) -- pair with
`rocprofv3 --kernel-trace` durations of block_mapped_kernel / chunk_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
src = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
for _ in range(3):
    _, st = ea.sssp(ctx, g, src, w, ea.Options(collect_kernel_time=True))
print("enact ms", st.elapsed_ms, "kernels ms", st.advance_kernel_ms, "iterations", st.iterations)
print("frontier slots", st.frontier_slots[:16])
print("edges expanded", st.edges_expanded, "= %.2f x E" % (st.edges_expanded / g.nnz), "edges traversed", st.edges_traversed)
