#!/usr/bin/env python3
"""Full-size check of the vertex-partitioned traversal: W ranks (gloo, sharing the visible GPU)
run BFS + SSSP on an R-MAT graph; every rank compares its label replica bit for bit with the
single-GPU engine's result on the whole graph.
  python -m torch.distributed.run --nproc-per-node W --master-addr 127.0.0.1 tools/verify_partitioned.py SCALE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import torch.distributed as dist
import essentials_amd as ea
from essentials_amd import api
from essentials_amd.distributed import HipKernels, PartitionedTraversal, OP_BFS, OP_SSSP

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
sources = ([int(x) for x in sys.argv[2].split(",")]
           if len(sys.argv) > 2 and sys.argv[2] != "soak" else [0, 12345])
# optional soak: verify_partitioned.py SCALE soak SECONDS SEED -- random sources, slot sizes and
# dense-exchange thresholds (the same draws on every rank) until the time is up
soak = len(sys.argv) > 2 and sys.argv[2] == "soak"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
ctx = ea.Context(0, stream=stream.cuda_stream)
full = ea.Graph.rmat(ctx, scale, 16, 1, 7)
h, lo, hi = api._VP(), C.c_int32(), C.c_int32()
api._check(api.load_library().grx_graph_partition(full._h, rank, world, C.byref(h), C.byref(lo),
                                                  C.byref(hi)), "partition")
local = ea.Graph(h)
with torch.cuda.stream(stream):
    trav = PartitionedTraversal(HipKernels(ctx, local), dist, rank, world, full.n_rows, lo.value,
                                hi.value, local.nnz, "cuda:0", stream=stream)
bad = 0
if soak:
    import time
    import numpy as np
    budget, rng = float(sys.argv[3]), np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    n, runs, t0 = full.n_rows, 0, time.time()
    go = torch.ones(1)
    while True:
        go[0] = 1.0 if time.time() - t0 < budget else 0.0
        dist.broadcast(go, 0)           # rank 0's clock decides for everybody
        if not bool(go[0]):
            break
        kw = dict(small_slot=int(rng.choice([8, 512, 1 << 15])),
                  dense_threshold=int(rng.choice([0, 64, n // 64, 1 << 30])),
                  replica_threshold=int(rng.choice([0, 64, n // max(world, 1), 1 << 30])))
        with torch.cuda.stream(stream):
            t = PartitionedTraversal(HipKernels(ctx, local), dist, rank, world, n, lo.value, hi.value,
                                     local.nnz, "cuda:0", stream=stream, **kw)
        for s in rng.integers(0, n, 3):
            s = int(s)
            want_d, _ = ea.bfs(ctx, full, s)
            want_w, _ = ea.sssp(ctx, full, s)
            ctx.synchronize()
            depth = torch.empty(n, dtype=torch.int32, device="cuda"); t.run(OP_BFS, s, depth)
            w = torch.empty(n, dtype=torch.float32, device="cuda"); t.run(OP_SSSP, s, w)
            torch.cuda.synchronize()
            ok = torch.equal(depth, want_d) and torch.equal(w.view(torch.int32), want_w.view(torch.int32))
            runs += 1
            if not ok:
                bad += 1
                print(f"[rank {rank}] MISMATCH source {s} {kw}", flush=True)
    print(f"[rank {rank}/{world}] soak scale {scale}: {runs} sources, {bad} mismatches", flush=True)
    sources = []
for s in sources:
    want_d, _ = ea.bfs(ctx, full, s)
    want_w, _ = ea.sssp(ctx, full, s)
    ctx.synchronize()
    depth = torch.empty(full.n_rows, dtype=torch.int32, device="cuda")
    st = trav.run(OP_BFS, s, depth)
    w = torch.empty(full.n_rows, dtype=torch.float32, device="cuda")
    st2 = trav.run(OP_SSSP, s, w)
    torch.cuda.synchronize()
    ok_d = bool(torch.equal(depth, want_d))
    ok_w = bool(torch.equal(w.view(torch.int32), want_w.view(torch.int32)))
    bad += (not ok_d) + (not ok_w)
    print(f"[rank {rank}/{world}] scale {scale} source {s}: bfs {'OK' if ok_d else 'MISMATCH'} "
          f"({st['supersteps']} supersteps, {st['bitmap_supersteps']} bitmap), "
          f"sssp {'OK' if ok_w else 'MISMATCH'} ({st2['supersteps']} supersteps, "
          f"{st2.get('allreduce_supersteps', 0)} all-reduce); rows {lo.value}..{hi.value}",
          flush=True)
t = torch.tensor([bad])
dist.all_reduce(t)
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if int(t.item()) else 0)
