#!/usr/bin/env python3
"""Push BFS with one byte per vertex while it runs against the caller's 4-byte depths
(GRX_BFS_BYTE_LABELS): mean enact() and advance-kernel time of a few sources.  usage: [scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
scale = int(sys.argv[1])
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
deg = np.diff(g.offsets_to_host())
rng = np.random.default_rng(3)
srcs = [0] + rng.choice(np.flatnonzero(deg > 0), 3).tolist()
ref = {}
for by in ("1", "0"):
    os.environ["GRX_BFS_BYTE_LABELS"] = by
    te = tk = 0.0
    for s in srcs:
        best = None
        for _ in range(3):
            _, st = ea.bfs(ctx, g, int(s), d, ea.Options(collect_kernel_time=True))
            if best is None or st.elapsed_ms < best.elapsed_ms:
                best = st
        te += best.elapsed_ms
        tk += best.advance_kernel_ms
        h = int(torch.sum(torch.where(d < 2**31 - 1, d, torch.zeros_like(d)).long()).item())
        assert ref.setdefault(s, h) == h, (s, by)
    print(f"scale {scale} byte_labels={by}: mean enact {te / len(srcs):.3f} ms, advance kernels {tk / len(srcs):.3f} ms", flush=True)
