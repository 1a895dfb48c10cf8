#!/usr/bin/env python3
"""Upper bound of what a workgroup-local write-combining table could do for the push PageRank
(VERDICT r2 item 5): the bare scatter acc[dst] += w over the directed R-MAT graph with the edges into
the K HOTTEST destinations taken out of the atomic stream (redirected to uniformly random cold
destinations: an ideal K-entry combiner absorbs exactly those, the rest still goes to memory).
usage: pr_hot_probe.py [scale]          (on the GPU box)
       pr_hot_probe.py --simulate       (CPU, numpy): what a FIRST-COME direct-mapped table of S slots
                                        absorbs of one workgroup's share of the edges (destinations
                                        drawn from the R-MAT column law, every id bit 1 with p = 0.24)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if "--simulate" in sys.argv:
    rng = np.random.default_rng(1)
    scale = 24
    for share in (262144, 1048576):
        d = ((rng.random((share, scale)) < 0.24) * (1 << np.arange(scale))).sum(1).astype(np.int64)
        for slots in (1024, 2048, 4096, 8192, 16384):
            h = (d * 2654435761 % (1 << 32)) >> (32 - int(np.log2(slots)))
            _, first = np.unique(h, return_index=True)       # the first comer owns the slot
            owner = np.full(slots, -1, np.int64)
            owner[h[first]] = d[first]
            absorbed = int((owner[h] == d).sum()) - len(first)
            print(f"workgroup share {share} edges, {slots} slots: {absorbed / share:.1%} absorbed")
    sys.exit(0)

import torch
import essentials_amd as ea

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 0, symmetrize=False)
V, E = g.n_rows, g.nnz
Ap, Aj, Ax = g.to_host()
ap = torch.from_numpy(Ap).cuda()
aj = torch.from_numpy(np.ascontiguousarray(Aj)).cuda()
val = torch.ones(E, dtype=torch.float32, device="cuda")
indeg = torch.bincount(aj.long(), minlength=V)
order = torch.argsort(indeg, descending=True)
acc = torch.zeros(V, dtype=torch.float32, device="cuda")
print(f"directed R-MAT scale {scale}: V={V} E={E}, hottest destination receives {int(indeg.max())} edges", flush=True)


def scatter(graph):
    best = 1e9
    for _ in range(3):
        acc.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ea.advance(ctx, graph, None, ea.EdgeOp.sum_weight, acc, 0, want_output=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for K in (0, 16, 256, 2048, 16384, 131072):
    hot = torch.zeros(V, dtype=torch.bool, device="cuda")
    hot[order[:K]] = True
    into_hot = hot[aj.long()]
    share = float(into_hot.float().mean())
    col = aj.clone()
    n_hot = int(into_hot.sum())
    if n_hot:
        col[into_hot] = torch.randint(0, V, (n_hot,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gk = ea.Graph.from_device_csr(ap, col, val)
    t = scatter(gk)
    # an ideal combiner would not issue the absorbed share at all: scale the time of the rest
    print(f"K = {K:7d} hottest destinations out of the stream ({share:6.1%} of the edges): all {E/1e6:.0f} M atomics "
          f"{t*1e3:7.2f} ms = {E/t/1e9:5.2f} G/s; without the absorbed share ~{t*(1-share)*1e3:6.2f} ms", flush=True)
    gk.close()
