#!/usr/bin/env python3
"""What a degree-ordered vertex numbering is worth: the same R-MAT graph (i) as generated (low ids
are the hot ones by construction: no scramble), (ii) with vertex ids in descending-degree order,
(iii) with randomly scrambled ids (Graph500 style), through the unchanged engine.
BFS depths / SSSP distances are compared through the permutation.

usage: reorder_probe.py [scale] [n_sources]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n_src = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
Ap = g.offsets_to_host()
n = g.n_rows
dev = 'cuda'
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from layout_lib import relabelled

rng = np.random.default_rng(5)
cand = np.flatnonzero(np.diff(Ap) > 0)
sources = [0] + [int(x) for x in rng.choice(cand, n_src - 1, replace=False)]

layouts = ("generated", "degree-ordered", "scrambled")
ref_depth, ref_dist = {}, {}
for name in layouts:
    gg, rank = relabelled(ctx, g, name)
    d = torch.empty(n, dtype=torch.int32, device=dev)
    w = torch.empty(n, dtype=torch.float32, device=dev)
    for every_edge in (False, True):
        o = ea.Options(collect_kernel_time=True, call_every_edge=every_edge)
        be, bk, se, sk = [], [], [], []
        for s in sources:
            s2 = s if rank is None else int(rank[s])
            for _ in range(2):
                _, st = ea.bfs(ctx, gg, s2, d, o)
            be.append(st.elapsed_ms); bk.append(st.advance_kernel_ms)
            for _ in range(2):
                _, st2 = ea.sssp(ctx, gg, s2, w, o)
            se.append(st2.elapsed_ms); sk.append(st2.advance_kernel_ms)
            dd = d if rank is None else d[rank]
            ww = w if rank is None else w[rank]
            if name == "generated" and not every_edge:
                ref_depth[s], ref_dist[s] = dd.clone(), ww.clone()
            else:
                assert torch.equal(dd, ref_depth[s]), (name, s, "bfs")
                assert torch.equal(ww.view(torch.int32), ref_dist[s].view(torch.int32)), (name, s, "sssp")
        print(f"{name:15s} every_edge={int(every_edge)}: BFS enact {np.mean(be):.3f} ms kernels {np.mean(bk):.3f} | "
              f"SSSP enact {np.mean(se):.3f} ms kernels {np.mean(sk):.3f}   (mean of {len(sources)} sources; "
              f"source 0: BFS {bk[0]:.3f} SSSP {sk[0]:.3f})", flush=True)
    if rank is not None:
        gg.close()
