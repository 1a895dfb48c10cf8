"""Per-iteration work of the SSSP client (run with GRX_DEBUG=1: each advance prints its output size
and the degree sum of what it emitted = the edges the next iteration relaxes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, essentials_amd as ea
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, 22, 16, 1, 7)
for s in (0, 12345):
    print("== source", s, flush=True)
    d, st = ea.sssp(ctx, g, s)
    print("iters", st.iterations, "enact", st.elapsed_ms, "slots", st.frontier_slots[:12], flush=True)
