import os, sys
sys.path.insert(0, "/root/repo")
import torch, essentials_amd as ea
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, 22, 16, 1, 7)
for s in (0, 12345):
    print("== source", s, flush=True)
    d, st = ea.sssp(ctx, g, s)
    print("iters", st.iterations, "enact", st.elapsed_ms, "slots", st.frontier_slots[:12], flush=True)
