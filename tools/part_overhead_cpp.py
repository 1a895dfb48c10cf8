"""Fixed cost of the partitioned protocol: a job of ONE rank through grx_partitioned_run (the C++
superstep loop: step -> gather -> host wait per superstep; RCCL when GRX_RCCL=1, else no transport)
against grx_bfs / grx_sssp on the same graph, with and without hot-first numbering.
usage: part_overhead_cpp.py [scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import essentials_amd as ea
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = ea.Context(0)
if os.environ.get("GRX_RCCL", "1") == "1":
    ctx.attach_rccl(0, 1, ea.Context.unique_id())
single = ea.Context(0)   # grx_bfs / grx_sssp on a context of their own (no job attached)
g = ea.Graph.rmat(ctx, scale, 16, 1, 7)
import ctypes as C
from essentials_amd import api
if os.environ.get("GRX_HOT_FIRST", "1") != "0":   # what bench.py's N > 1 runner does
    h, lo, hi = api._VP(), C.c_int32(), C.c_int32()
    api._check(api.load_library().grx_graph_partition_hot_first(ctx._h, g._h, 0, 1, C.byref(h), C.byref(lo), C.byref(hi)), "partition")
    local = ea.Graph(h)
    plan = ea.PartitionedPlan(ctx, local, lo.value, hi.value)
else:
    plan = ea.PartitionedPlan(ctx, g, 0, g.n_rows)
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
d2, w2 = d.clone(), w.clone()
rng = np.random.default_rng(3)
deg = np.diff(g.offsets_to_host())
sources = [0] + [int(s) for s in rng.choice(np.flatnonzero(deg > 0), 5, replace=False)]
rows = []
for s in sources:
    for _ in range(2):
        st = plan.run(ea.EdgeOp.bfs, s, d); st2 = plan.run(ea.EdgeOp.sssp, s, w)
        _, a = ea.bfs(single, g, s, d2); _, b = ea.sssp(single, g, s, w2)
        _, a0 = ea.bfs(single, g, s, d2, ea.Options(call_every_edge=True))
    assert torch.equal(d, d2) and torch.equal(w.view(torch.int32), w2.view(torch.int32))
    rows.append((st["elapsed_ms"], a.elapsed_ms, a0.elapsed_ms, st2["elapsed_ms"], b.elapsed_ms))
    print(f"source {s}: partitioned(world 1, {ctx.job_info()['backend']}) bfs {st['elapsed_ms']:.3f} ms / {st['supersteps']} supersteps "
          f"({st['collectives']} collectives) | grx_bfs {a.elapsed_ms:.3f} (every edge, caller's numbering {a0.elapsed_ms:.3f}) || "
          f"sssp {st2['elapsed_ms']:.3f} ms / {st2['supersteps']} ({st2['collectives']}) | grx_sssp {b.elapsed_ms:.3f}", flush=True)
m = np.mean(rows, 0)
print(f"mean of {len(sources)} sources: partitioned bfs {m[0]:.3f} vs grx_bfs {m[1]:.3f} (every-edge {m[2]:.3f}); partitioned sssp {m[3]:.3f} vs grx_sssp {m[4]:.3f}")
