#!/usr/bin/env python3
"""Compare the advance schedules on one R-MAT graph: BFS / SSSP enact time and kernel time."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import essentials_amd as ea

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--lbs", default="block_mapped,work_stealing,merge_path,bucketing,warp_mapped")
ap.add_argument("--algo", default="bfs,sssp")
ap.add_argument("--source", type=int, default=0)
ap.add_argument("--hub", default="0")
ap.add_argument("--chunk", default="0")
ap.add_argument("--sorted", action="store_true")
a = ap.parse_args()
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, a.scale, 16, 1, 7)
if a.sorted:
    g = g.sorted_rows(ctx)
print(f"graph: V={g.n_rows} E={g.nnz} sorted_rows={a.sorted}", flush=True)
print("copy roof GB/s:", round(ctx.copy_bandwidth_gbps(1 << 30, 10), 1), flush=True)
for algo in a.algo.split(","):
  fn = ea.bfs if algo == "bfs" else ea.sssp
  for lb in a.lbs.split(","):
    for hub in [int(x) for x in a.hub.split(",")]:
      for chunk in [int(x) for x in a.chunk.split(",")]:
        best = None
        for r in range(a.reps):
            _, st = fn(ctx, g, a.source, options=ea.Options(load_balance=ea.LoadBalance[lb], collect_kernel_time=True, hub_threshold=hub, chunk_edges=chunk))
            if best is None or st.elapsed_ms < best.elapsed_ms:
                best = st
        st2 = None
        for r in range(a.reps):
            _, st = fn(ctx, g, a.source, options=ea.Options(load_balance=ea.LoadBalance[lb], hub_threshold=hub, chunk_edges=chunk))
            if st2 is None or st.elapsed_ms < st2.elapsed_ms:
                st2 = st
        print(f"{algo:5s} {lb:14s} hub {hub:5d} chunk {chunk:5d} enact {st2.elapsed_ms:7.3f} ms (timed-kernels run {best.elapsed_ms:7.3f})  kernels {best.advance_kernel_ms:7.3f} ms "
              f"iters {best.iterations} GTEPS(enact) {best.edges_traversed/st2.elapsed_ms/1e6:7.1f} GTEPS(kernel) {best.edges_traversed/best.advance_kernel_ms/1e6:7.1f}", flush=True)
