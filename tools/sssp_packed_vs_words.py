import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import essentials_amd as ea
scale = int(sys.argv[1])
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
deg = np.diff(g.offsets_to_host())
rng = np.random.default_rng(3)
srcs = [0] + rng.choice(np.flatnonzero(deg > 0), 2).tolist()
ref = {}
for pk in ("1", "0"):
    os.environ["GRX_SSSP_PACKED"] = pk
    tot = 0.0
    for s in srcs:
        best = None
        for _ in range(3):
            _, st = ea.sssp(ctx, g, int(s), w)
            best = st.elapsed_ms if best is None else min(best, st.elapsed_ms)
        tot += best
        h = int(torch.sum(w[w < 3e38].double()).item() * 16)
        assert ref.setdefault(s, h) == h, (s, pk)
    print(f"scale {scale} packed={pk}: mean enact {tot / len(srcs):.2f} ms over {len(srcs)} sources", flush=True)
