#!/usr/bin/env python3
"""Why is the PUSH PageRank (one float atomicAdd per edge, reference algorithms/pr.hxx:140-146) slow?
The same graph -> none advance with the bare scatter functor (GRX_OP_SUM_WEIGHT: acc[dst] += w) on
  (a) the directed R-MAT graph (skewed destinations: hubs),
  (b) the same rows with UNIFORM random destinations (no hubs, same edge count),
  (c) the R-MAT graph with column-sorted rows,
and the PageRank push / pull iterations on (a) and (c).  Pair with
`rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_BUSY_sum TCC_CYCLE_sum` for the atomic counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 0, symmetrize=False)
V, E = g.n_rows, g.nnz
print(f"graph V={V} E={E} (directed R-MAT scale {scale})", flush=True)
ap = torch.from_numpy(g.offsets_to_host()).cuda()
col = torch.randint(0, V, (E,), dtype=torch.int32, device="cuda")
val = torch.ones(E, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
uniform = ea.Graph.from_device_csr(ap, col, val)
srt = g.sorted_rows(ctx)
acc = torch.zeros(V, dtype=torch.float32, device="cuda")


def scatter(graph, name):
    best = 1e9
    for _ in range(4):
        acc.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ea.advance(ctx, graph, None, ea.EdgeOp.sum_weight, acc, 0, want_output=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"scatter acc[dst] += w  {name:32s} {best*1e3:8.2f} ms  {E/best/1e9:6.2f} G atomics/s "
          f"(sum {float(acc.sum()):.0f})", flush=True)


scatter(g, "R-MAT destinations")
scatter(uniform, "uniform random destinations")
scatter(srt, "R-MAT, column-sorted rows")
for graph, name in ((g, "R-MAT"), (srt, "R-MAT, column-sorted rows")):
    best = None
    for _ in range(2):
        _, st = ea.pagerank(ctx, graph, 0.85, 1e-6)
        if best is None or st.elapsed_ms < best.elapsed_ms:
            best = st
    print(f"pagerank push {name:28s} {best.elapsed_ms/best.iterations:8.2f} ms/iteration ({best.iterations} iterations)",
          flush=True)
g.build_in_edges(ctx)
_, st = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
print(f"pagerank pull R-MAT {'':22s} {st.elapsed_ms/st.iterations:8.2f} ms/iteration ({st.iterations} iterations)")
