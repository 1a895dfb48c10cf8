#!/usr/bin/env python3
"""Per-level work of one push BFS (edges expanded in place by the tile kernel vs. cut into hub
chunks) -- pair with `rocprofv3 --kernel-trace` durations of block_mapped_kernel / chunk_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
src = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
hub = int(os.environ.get("HUB", "0"))
for _ in range(3):
    _, st = ea.bfs(ctx, g, src, d, ea.Options(collect_kernel_time=True, hub_threshold=hub))
print("enact ms", st.elapsed_ms, "kernels ms", st.advance_kernel_ms, "slots", st.frontier_slots[:10])
deg = np.diff(g.to_host()[0]).astype(np.int64)
depth = d.cpu().numpy()
hub_t = hub or 256
for L in range(int(depth[depth < 2**31 - 1].max()) + 1):
    m = depth == L
    dl = deg[m]
    hub = dl[dl >= (hub_t)]
    print(f"level {L}: {m.sum():8d} vertices, tile edges {dl[dl < hub_t].sum():10d} (max tile-vertex deg {dl[dl<hub_t].max() if (dl<hub_t).any() else 0}), hub edges {hub.sum():10d} in {len(hub)} hubs")
