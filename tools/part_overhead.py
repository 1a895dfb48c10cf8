"""Fixed cost of the partitioned superstep protocol: world=1 PartitionedTraversal vs grx_bfs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import essentials_amd as ea
from essentials_amd.distributed import HipKernels, PartitionedTraversal, OP_BFS, OP_SSSP
stream = torch.cuda.Stream()
ctx = ea.Context(0, stream=stream.cuda_stream)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
fused = os.environ.get("GRX_FUSED", "1") == "1"
trav = PartitionedTraversal(HipKernels(ctx, g), None, 0, 1, g.n_rows, 0, g.n_rows, g.nnz, "cuda:0", fused=fused, stream=stream)
print("fused", fused)
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
for i in range(3):
    st = trav.run(OP_BFS, 0, d); st2 = trav.run(OP_SSSP, 0, w)
    _, a = ea.bfs(ctx, g, 0); _, b = ea.sssp(ctx, g, 0)
    print(f"partitioned(world=1) bfs {st['elapsed_ms']:.3f} ms / {st['supersteps']} supersteps, sssp {st2['elapsed_ms']:.3f} ms / {st2['supersteps']}"
          f" | single bfs {a.elapsed_ms:.3f} sssp {b.elapsed_ms:.3f}", flush=True)
    print("   bfs", {k: v for k, v in st.items() if k != "elapsed_ms"})
