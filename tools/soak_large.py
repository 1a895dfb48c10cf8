#!/usr/bin/env python3
"""Large-graph soak: random sources on an R-MAT graph; push BFS (several schedules), the
direction-optimising BFS and the partitioned protocol (world 1) must give identical depths, SSSP
identical distance bits across schedules and the protocol.  usage: soak_large.py [scale] [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
from essentials_amd.distributed import HipKernels, PartitionedTraversal, OP_BFS, OP_SSSP
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
stream = torch.cuda.Stream()
ctx = ea.Context(0, stream=stream.cuda_stream)
g = ea.Graph.rmat(ctx, scale, 16, int(rng.integers(1, 1000)), 7)
n = g.n_rows
trav = PartitionedTraversal(HipKernels(ctx, g), None, 0, 1, n, 0, n, g.nnz, "cuda:0", stream=stream,
                            replica_threshold=n // 8)
lbs = ["block_mapped", "work_stealing", "merge_path", "bucketing"]
t0, runs, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    s = int(rng.integers(0, n))
    ref, st = ea.bfs(ctx, g, s)
    ref = ref.clone()
    wref, _ = ea.sssp(ctx, g, s)
    wref = wref.clone()
    lb = ea.LoadBalance[lbs[int(rng.integers(0, len(lbs)))]]
    o = ea.Options(load_balance=lb, hub_threshold=int(rng.choice([0, 64, 1024])))
    d, _ = ea.bfs(ctx, g, s, options=o)
    ok = torch.equal(d, ref)
    d, _ = ea.bfs(ctx, g, s, options=ea.Options(direction_optimized=True, do_alpha=float(rng.choice([1, 4, 16]))))
    ok &= torch.equal(d, ref)
    w, _ = ea.sssp(ctx, g, s, options=o)
    ok &= torch.equal(w.view(torch.int32), wref.view(torch.int32))
    # the forms that stand for the unchanged reference clients: caller's numbering, every edge / two passes
    d, _ = ea.bfs(ctx, g, s, options=ea.Options(call_every_edge=True))
    ok &= torch.equal(d, ref)
    w, _ = ea.sssp(ctx, g, s, options=ea.Options(sssp_two_pass=True))
    ok &= torch.equal(w.view(torch.int32), wref.view(torch.int32))
    dp = torch.empty(n, dtype=torch.int32, device="cuda"); trav.run(OP_BFS, s, dp)
    ok &= torch.equal(dp, ref)
    wp = torch.empty(n, dtype=torch.float32, device="cuda"); trav.run(OP_SSSP, s, wp)
    ok &= torch.equal(wp.view(torch.int32), wref.view(torch.int32))
    runs += 1
    if not ok:
        bad += 1
        print("MISMATCH source", s, lb.name, flush=True)
print(f"scale {scale}: {runs} sources, {bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
