"""Full-size runs of what only bench.py exercised before (missing item 4): graphs beyond
2^22 vertices, where the label forms switch BY THEMSELVES (one byte per vertex in the push BFS, two
words instead of the packed label in SSSP) and the hot-first renumbered copy is the default, and
BASELINE configs[3]'s stand-in (PageRank on a directed R-MAT-24).  No oracle at these sizes:
size-independent certificates, computed on the device with torch ops from the graph's arrays.
"""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

INF_I = 2**31 - 1
INF_F = float(np.finfo(np.float32).max)


@pytest.fixture(scope="module")
def ea():
    import essentials_amd
    return essentials_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


@pytest.fixture(scope="module")
def ctx(ea, torch):
    return ea.Context(0)


def _device_csr(torch, g):
    h_ap, h_aj, h_ax = g.to_host()
    ap = torch.from_numpy(h_ap).cuda()
    aj = torch.from_numpy(np.ascontiguousarray(h_aj)).cuda()
    ax = torch.from_numpy(np.ascontiguousarray(h_ax)).cuda()
    return ap, aj, ax


def test_bfs_sssp_rmat24_default_label_forms(ea, ctx, torch, monkeypatch):
    """RMAT-24 (16.8 M vertices, 537 M directed edges, symmetric): byte depths and two-word SSSP
    labels switch on by default at this size, on the hot-first copy.  (1) the labels equal those
    of the other label form and of the caller's numbering; (2) BFS and shortest-path certificates
    on the device (model: test_gpu_parity.py::test_bfs_rmat22_properties / test_sssp_...)."""
    g = ea.Graph.rmat(ctx, 24, 16, seed=1, weight_seed=7)
    n = g.n_rows
    assert n == 1 << 24 and n > 1 << 22          # beyond both default switches
    src_vertex = 0
    d0, st0 = ea.bfs(ctx, g, src_vertex)          # default: byte depths, hot-first copy
    w0, sw0 = ea.sssp(ctx, g, src_vertex)         # default: two-word labels, hot-first copy
    # the other label forms, and the caller's own numbering, give the same labels
    monkeypatch.setenv("GRX_BFS_BYTE_LABELS", "0")
    d1, st1 = ea.bfs(ctx, g, src_vertex)
    monkeypatch.delenv("GRX_BFS_BYTE_LABELS")
    monkeypatch.setenv("GRX_SSSP_PACKED", "1")
    w1, _ = ea.sssp(ctx, g, src_vertex)
    monkeypatch.delenv("GRX_SSSP_PACKED")
    assert torch.equal(d0, d1) and st0.frontier_slots == st1.frontier_slots
    assert torch.equal(w0.view(torch.int32), w1.view(torch.int32))
    del d1, w1
    g.hot_first(ctx, False)
    d2, st2 = ea.bfs(ctx, g, src_vertex)
    w2, _ = ea.sssp(ctx, g, src_vertex)
    assert torch.equal(d0, d2) and st0.edges_traversed == st2.edges_traversed
    assert torch.equal(w0.view(torch.int32), w2.view(torch.int32))
    del d2, w2
    # certificates
    ap, aj, ax = _device_csr(torch, g)
    deg = (ap[1:] - ap[:-1]).long()
    u = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int32), deg).long()
    v = aj.long()
    du, dv = d0[u], d0[v]
    reached_u = du != INF_I
    assert int(d0[src_vertex]) == 0
    assert bool((dv[reached_u] != INF_I).all())                        # closed under edges
    assert bool((dv[reached_u] <= du[reached_u] + 1).all())             # no edge skips a level
    best = torch.full((n,), INF_I, dtype=torch.int32, device="cuda")
    best.scatter_reduce_(0, v, du, reduce="amin")                       # min depth over in-neighbours
    reached = d0 != INF_I
    nonsrc = reached.clone()
    nonsrc[src_vertex] = False
    assert bool((best[nonsrc] + 1 == d0[nonsrc]).all())                 # tight: a parent exists
    assert st0.vertices_reached == int(reached.sum())
    assert st0.edges_traversed == int(deg[reached].sum())
    del du, dv, best
    wu, wv = w0[u], w0[v]
    reached_u = wu < INF_F
    assert float(w0[src_vertex]) == 0.0
    assert bool((wv[reached_u] < INF_F).all())
    assert bool((wv[reached_u] <= wu[reached_u] + ax[reached_u]).all())  # no edge can still relax
    cand = torch.where(reached_u, wu + ax, torch.full_like(wu, INF_F))
    bestw = torch.full((n,), INF_F, dtype=torch.float32, device="cuda")
    bestw.scatter_reduce_(0, v, cand, reduce="amin")
    reached = w0 < INF_F
    nonsrc = reached.clone()
    nonsrc[src_vertex] = False
    assert bool((bestw[nonsrc] == w0[nonsrc]).all())                     # tight: a parent exists
    assert sw0.vertices_reached == int(reached.sum())
    g.close()


def test_pagerank_rmat24_directed_properties(ea, ctx, torch):
    """BASELINE configs[3] stand-in at full size (directed R-MAT-24, 268 M edges; the reference has
    no checker for pr.hxx, examples/algorithms/pr/pr.cu:64-70 -- parity unpinned): the ranks sum
    to 1 within 1e-3, one more iteration moves no rank by more than the tolerance, and the push
    form (pr.hxx's scatter) and the pull form agree within 5e-6."""
    g = ea.Graph.rmat(ctx, 24, 16, 1, 0, False)
    tol = 1e-6
    p, st = ea.pagerank(ctx, g, 0.85, tol)
    assert st.iterations >= 2
    assert abs(float(p.double().sum()) - 1.0) < 1e-3
    assert float(p.min()) > 0.0
    p_next, st_next = ea.pagerank(ctx, g, 0.85, tol, options=ea.Options(max_iterations=st.iterations + 1))
    # max_iterations only ever stops EARLIER than the tolerance would: the longer run converged too,
    # or stopped one iteration later -- either way no rank moved by more than the tolerance
    assert st_next.iterations in (st.iterations, st.iterations + 1)
    assert float((p_next - p).abs().max()) <= tol
    g.build_in_edges(ctx)
    q, sq = ea.pagerank(ctx, g, 0.85, tol, options=ea.Options(direction_optimized=True))
    assert abs(sq.iterations - st.iterations) <= 1
    assert float((q - p).abs().max()) < 5e-6
    g.close()


def _rank_rmat26(rank, world, port, out_dir):
    """One of three ranks sharing the GPU (gloo; RCCL refuses two ranks on one device): the C++
    superstep loop over host callbacks on BASELINE configs[4]'s graph."""
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import essentials_amd as ea
    from essentials_amd import api
    from essentials_amd.distributed import attach_job
    torch.cuda.set_device(0)
    ctx = ea.Context(0)
    full = ea.Graph.rmat(ctx, 26, 16, 1, 7)
    h, lo, hi = api._VP(), C.c_int32(), C.c_int32()
    api._check(api.load_library().grx_graph_partition(full._h, rank, world, C.byref(h), C.byref(lo),
                                                      C.byref(hi)), "partition")
    local = ea.Graph(h)
    n = full.n_rows
    notes = []
    want = {}
    if rank == 0:   # the single-GPU engine's labels (one rank computes them: memory)
        single = ea.Context(0)
        for s in (0, 777777):
            d, _ = ea.bfs(single, full, s)
            w, _ = ea.sssp(single, full, s)
            want[s] = (d.cpu(), w.view(torch.int32).cpu())
            del d, w
        single.close()
        ea.Context.trim_cache()
    full.close()
    dist.barrier()
    attach_job(ctx, dist)
    plan = ea.PartitionedPlan(ctx, local, lo.value, hi.value)
    for s in (0, 777777):
        depth = torch.empty(n, dtype=torch.int32, device="cuda")
        st = plan.run(ea.EdgeOp.bfs, s, depth)
        w = torch.empty(n, dtype=torch.float32, device="cuda")
        st2 = plan.run(ea.EdgeOp.sssp, s, w)
        mine = (depth.cpu(), w.view(torch.int32).cpu())
        del depth, w
        ref = [want.get(s)]
        dist.broadcast_object_list(ref, src=0)
        if not torch.equal(mine[0], ref[0][0]):
            notes.append(f"bfs {s}: depths differ from the single-GPU engine's {st}")
        if not torch.equal(mine[1], ref[0][1]):
            notes.append(f"sssp {s}: distances differ from the single-GPU engine's {st2}")
        if st["collectives"] != st["supersteps"] + st["bitmap_supersteps"] + st["large_gather_supersteps"]:
            notes.append(f"bfs {s}: collective count {st}")
        if rank == 0:
            print(f"RMAT-26 x{world} source {s}: bfs {st['supersteps']} supersteps ({st['bitmap_supersteps']} "
                  f"bitmap) {st['elapsed_ms']:.0f} ms; sssp {st2['supersteps']} supersteps "
                  f"({st2['allreduce_supersteps']} all-reduce) {st2['elapsed_ms']:.0f} ms", flush=True)
    open(os.path.join(out_dir, f"rank{rank}." + ("bad" if notes else "ok")), "w").write(str(notes))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_three_ranks_rmat26(tmp_path):
    """BASELINE configs[4]'s graph (R-MAT scale 26, 2.15 G directed edges -- still 32-bit edge ids),
    three ranks over the callback transport: every replica equals the single-GPU engine's labels."""
    import os
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 3
    mp.spawn(_rank_rmat26, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    names = sorted(os.listdir(tmp_path))
    notes = {f: open(os.path.join(tmp_path, f)).read() for f in names}
    assert names == [f"rank{r}.ok" for r in range(world)], notes
