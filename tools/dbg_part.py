import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import essentials_amd as ea
from essentials_amd.distributed import HipKernels, PartitionedTraversal, OP_BFS, OP_SSSP
from oracle.oracle import Oracle
o = Oracle()
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, 12, 16, 1, 7)
Ap, Aj, Ax = g.to_host()
for lb in (ea.LoadBalance.block_mapped, ea.LoadBalance.merge_path, ea.LoadBalance.bucketing):
    for ss in (None, 64):
        trav = PartitionedTraversal(HipKernels(ctx, g, ea.Options(load_balance=lb)), None, 0, 1, g.n_rows, 0, g.n_rows, g.nnz, "cuda:0", small_slot=ss)
        for s in (0, 1830):
            depth = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
            st = trav.run(OP_BFS, s, depth)
            want, _ = o.bfs_heap(Ap, Aj, s)
            bad = int((depth.cpu().numpy() != want).sum())
            d = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
            st2 = trav.run(OP_SSSP, s, d)
            wantw, _ = o.sssp_heap(Ap, Aj, Ax, s)
            bad2 = int((d.cpu().numpy().view(np.uint32) != wantw.view(np.uint32)).sum())
            print(lb.name, ss, s, "bfs wrong", bad, st, "sssp wrong", bad2, st2["supersteps"], flush=True)
