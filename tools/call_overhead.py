#!/usr/bin/env python3
"""Host cost of one grx_bfs / grx_sssp call beside the device time enact() reports: wall time of
the whole Python call minus Stats.elapsed_ms, on RMAT-22 (the bench's step) and on a tiny graph
(where nearly everything is fixed cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import essentials_amd as ea

ctx = ea.Context(0)
for scale in (10, 22):
    g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
    d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
    w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
    for name, fn, buf in (("bfs", ea.bfs, d), ("sssp", ea.sssp, w)):
        for _ in range(3):
            fn(ctx, g, 0, buf)
        torch.cuda.synchronize()
        wall, dev = [], []
        for _ in range(20):
            t0 = time.perf_counter()
            _, st = fn(ctx, g, 0, buf)
            wall.append((time.perf_counter() - t0) * 1e3)
            dev.append(st.elapsed_ms)
        wall.sort(); dev.sort()
        print(f"scale {scale:2d} {name:4s}: call {wall[len(wall)//2]:.3f} ms, enact {dev[len(dev)//2]:.3f} ms, "
              f"outside enact {wall[len(wall)//2] - dev[len(dev)//2]:.3f} ms ({st.iterations} iterations)", flush=True)
    g.close()
