"""Randomised parity sweeps and the rarely-taken paths of the advance kernels: tiny hub
thresholds / chunk sizes, a capped chunk queue (overflow -> hubs expanded in place), output
frontiers that must grow, many sources.  Everything is checked against the oracle."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

INF_I = 2**31 - 1


@pytest.fixture(scope="module")
def env(oracle):
    import torch
    import essentials_amd as ea
    assert torch.cuda.is_available()
    return ea, ea.Context(0), torch


def host(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("lb", ["block_mapped", "bucketing", "work_stealing"])
def test_chunk_queue_overflow_falls_back_in_place(env, oracle, lb):
    ea, ctx, torch = env
    n, Ap, Aj, Ax = oracle.rmat_csr(13, 16, 3, 7)
    Aj = np.ascontiguousarray(Aj)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    want, _ = oracle.bfs_heap(Ap, Aj, 0)
    wantw, _ = oracle.sssp_heap(Ap, Aj, np.ascontiguousarray(Ax), 0)
    for hub, chunk, limit in ((8, 4, 3), (16, 16, 1), (64, 64, 50), (2, 1, 7), (256, 1024, 2)):
        o = ea.Options(load_balance=ea.LoadBalance[lb], hub_threshold=hub, chunk_edges=chunk,
                       chunk_queue_limit=limit)
        d, _ = ea.bfs(ctx, G, 0, options=o)
        assert (host(d) == want).all(), (lb, hub, chunk, limit)
        w, _ = ea.sssp(ctx, G, 0, options=o)
        assert (host(w).view(np.uint32) == wantw.view(np.uint32)).all(), (lb, hub, chunk, limit)
    # exactly-once per edge in the overflow path too
    f = torch.arange(n, dtype=torch.int32, device="cuda")
    calls = torch.zeros(len(Aj), dtype=torch.int32, device="cuda")
    out = ea.advance(ctx, G, f, ea.EdgeOp.count_edge, calls, 0,
                     ea.Options(load_balance=ea.LoadBalance[lb], hub_threshold=4, chunk_edges=2,
                                chunk_queue_limit=5), capacity=len(Aj) + 16)
    assert (host(calls) == 1).all()
    src = np.repeat(np.arange(n), np.diff(Ap))
    assert sorted(host(out).tolist()) == sorted(Aj[(src + Aj) % 3 == 0].tolist())


def test_output_frontier_grows_when_needed(env, oracle):
    """frontier_sizing_factor far too small: the engine must size the output exactly and grow it
    (reference: block_mapped.hxx:171-173 reserve)."""
    ea, ctx, torch = env
    n, Ap, Aj, Ax = oracle.rmat_csr(12, 16, 1, 7)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    want, _ = oracle.bfs_heap(Ap, np.ascontiguousarray(Aj), 0)
    for lb in ("block_mapped", "merge_path", "bucketing", "thread_mapped"):
        for holes in (False, True):
            d, _ = ea.bfs(ctx, G, 0, options=ea.Options(load_balance=ea.LoadBalance[lb],
                                                        frontier_sizing_factor=1e-6, holes_layout=holes))
            assert (host(d) == want).all(), (lb, holes)


def test_random_graphs_sources_schedules(env, oracle):
    ea, ctx, torch = env
    # GRX_STRESS_SEED / GRX_STRESS_TRIALS: longer soaks with other seeds (default: the fixed 12)
    rng = np.random.default_rng(int(os.environ.get("GRX_STRESS_SEED", "2026")))
    lbs = ["block_mapped", "merge_path", "bucketing", "work_stealing", "thread_mapped", "warp_mapped"]
    for trial in range(int(os.environ.get("GRX_STRESS_TRIALS", "12"))):
        scale = int(rng.integers(5, 15))
        ef = int(rng.integers(1, 24))
        sym = bool(rng.integers(0, 2))
        seed = int(rng.integers(1, 1 << 30))
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, ef, seed, 7, sym)
        Aj = np.ascontiguousarray(Aj); Ax = np.ascontiguousarray(Ax)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        for s in rng.integers(0, n, 2):
            want, _ = oracle.bfs_heap(Ap, Aj, int(s))
            wantw, _ = oracle.sssp_heap(Ap, Aj, Ax, int(s))
            lb = lbs[int(rng.integers(0, len(lbs)))]
            o = ea.Options(load_balance=ea.LoadBalance[lb], holes_layout=bool(rng.integers(0, 2)),
                           hub_threshold=int(rng.choice([2, 16, 256, 100000])),
                           chunk_edges=int(rng.choice([1, 64, 1024])))
            d, _ = ea.bfs(ctx, G, int(s), options=o)
            assert (host(d) == want).all(), (trial, lb, s)
            w, _ = ea.sssp(ctx, G, int(s), options=o)
            assert (host(w).view(np.uint32) == wantw.view(np.uint32)).all(), (trial, lb, s)
            if sym:
                d2, _ = ea.bfs(ctx, G, int(s), options=ea.Options(direction_optimized=True,
                                                                   do_alpha=float(rng.choice([0.5, 4, 1e6]))))
                assert (host(d2) == want).all(), (trial, "do", s)


def test_graph_rebuilt_in_the_same_memory(env, oracle):
    """Regression (found by a 200-trial soak): the engine remembered a graph's max degree by the
    address of its row offsets; a hub-less graph followed by a hub graph of the same shape in the
    same (freed and reused) memory then skipped the hub chunk kernel and lost edges.  Handles now
    carry their own max degree and queued chunks are always expanded."""
    ea, ctx, torch = env
    n = 1024
    # ring: every degree 1 -> no hubs
    ring_ap = np.arange(n + 1, dtype=np.int32)
    ring_aj = ((np.arange(n) + 1) % n).astype(np.int32)
    ones = np.ones(n, np.float32)
    # star with the same |V| and |E|: vertex 0 -> 1..1023, and 1 -> 0
    star_ap = np.concatenate([[0], np.full(1, n - 1), np.full(n - 1, n)]).astype(np.int32)
    star_aj = np.concatenate([np.arange(1, n), [0]]).astype(np.int32)
    for _ in range(4):   # alternate so that the allocator hands the same blocks back
        for Ap, Aj in ((ring_ap, ring_aj), (star_ap, star_aj)):
            G = ea.Graph.from_host_csr(Ap, Aj, ones)
            want, _ = oracle.bfs_heap(Ap, Aj, 0)
            for opts in (ea.Options(), ea.Options(load_balance=ea.LoadBalance.bucketing),
                         ea.Options(load_balance=ea.LoadBalance.work_stealing)):
                d, _ = ea.bfs(ctx, G, 0, options=opts)
                assert (host(d) == want).all()
            G.close()


def test_random_partitioned_protocol_world1(env, oracle):
    """The superstep protocol with random slot sizes and dense-exchange thresholds (pairs / level
    bitmaps / replica all-reduce, fused and two-call loops) on random graphs: world = 1 exercises
    every device kernel of the exchange; GRX_STRESS_TRIALS lengthens it."""
    ea, ctx0, torch = env
    from essentials_amd.distributed import HipKernels, PartitionedTraversal, OP_BFS, OP_SSSP
    rng = np.random.default_rng(int(os.environ.get("GRX_STRESS_SEED", "2026")) + 1)
    stream = torch.cuda.Stream()
    ctx = ea.Context(0, stream=stream.cuda_stream)
    for trial in range(max(4, int(os.environ.get("GRX_STRESS_TRIALS", "12")) // 3)):
        scale = int(rng.integers(5, 14))
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, int(rng.integers(1, 20)), int(rng.integers(1, 1 << 30)), 7,
                                        bool(rng.integers(0, 2)))
        Aj = np.ascontiguousarray(Aj); Ax = np.ascontiguousarray(Ax)
        g = ea.Graph.from_host_csr(Ap, Aj, Ax)
        fused = bool(rng.integers(0, 2))
        kw = dict(small_slot=int(rng.choice([2, 8, 64, 1 << 15])), fused=fused, stream=stream,
                  dense_threshold=int(rng.choice([0, 4, 64, 1 << 30])),
                  replica_threshold=int(rng.choice([0, 4, 64, 1 << 30])))
        trav = PartitionedTraversal(HipKernels(ctx, g), None, 0, 1, n, 0, n, g.nnz, "cuda:0", **kw)
        for s in rng.integers(0, n, 2):
            depth = torch.empty(n, dtype=torch.int32, device="cuda")
            trav.run(OP_BFS, int(s), depth)
            want, _ = oracle.bfs_heap(Ap, Aj, int(s))
            assert (depth.cpu().numpy() == want).all(), (trial, kw, s)
            w = torch.empty(n, dtype=torch.float32, device="cuda")
            trav.run(OP_SSSP, int(s), w)
            wantw, _ = oracle.sssp_heap(Ap, Aj, Ax, int(s))
            assert (w.cpu().numpy().view(np.uint32) == wantw.view(np.uint32)).all(), (trial, kw, s)
        g.close()


def test_random_pagerank_pull_and_push(env, oracle):
    """Pull and push PageRank against the oracle's pr.hxx restatement on random graphs (directed
    ones with their transpose attached): row groups, hub chunks and dangling vertices all occur."""
    ea, ctx, torch = env
    rng = np.random.default_rng(int(os.environ.get("GRX_STRESS_SEED", "2026")) + 2)
    for trial in range(max(3, int(os.environ.get("GRX_STRESS_TRIALS", "12")) // 4)):
        scale = int(rng.integers(4, 14))
        sym = bool(rng.integers(0, 2))
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, int(rng.integers(1, 40)), int(rng.integers(1, 1 << 30)),
                                        int(rng.integers(0, 9)), sym)
        Aj = np.ascontiguousarray(Aj); Ax = np.ascontiguousarray(Ax)
        g = ea.Graph.from_host_csr(Ap, Aj, Ax)
        if not sym:
            g.build_in_edges(ctx)
        want, it = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
        for pull in (True, False):
            p, st = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(direction_optimized=pull))
            p = host(p)
            assert np.abs(p - want).max() < 5e-6 and abs(st.iterations - it) <= 1, (trial, pull, scale, sym)
        g.close()


def test_settled_form_on_small_and_odd_graphs(oracle, monkeypatch):
    """The settled-destination form of the wide BFS levels (operators/settled.hxx), forced onto
    every level of small graphs whose sizes are not multiples of the bitmap's 128-bit groups, with
    tiny hub thresholds and a capped chunk queue; depths, counts and frontier lengths against the
    oracle and against the functor-per-edge form.  (The thresholds are read when a context is made.)"""
    import torch
    import essentials_amd as ea
    monkeypatch.setenv("GRX_SETTLED_MIN_WORK", "1")
    monkeypatch.setenv("GRX_FUSED_MIN_SLOTS", "64")
    ctx = ea.Context(0)
    rng = np.random.default_rng(2024 + int(os.environ.get("GRX_STRESS_SEED", "0")))
    for trial in range(40):
        n = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 500, 1000, 4097, 20011]))
        m = int(rng.integers(0, 12 * n + 1))
        rows = np.sort(rng.integers(0, n, m)).astype(np.int32)
        hubs = rng.random(m) < 0.3                      # a few heavy destinations / sources
        cols = np.where(hubs, rng.integers(0, max(1, n // 50), m), rng.integers(0, n, m)).astype(np.int32)
        Ap = np.zeros(n + 1, np.int32)
        np.add.at(Ap, rows + 1, 1)
        Ap = np.cumsum(Ap).astype(np.int32)
        Aj = np.ascontiguousarray(cols)
        Ax = np.ones(m, np.float32)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        deg = np.diff(Ap)
        for s in {0, int(rng.integers(0, n)), int(np.argmax(deg))}:
            want, _ = oracle.bfs_heap(Ap, Aj, s)
            o = dict(hub_threshold=int(rng.choice([0, 4, 64])), chunk_edges=int(rng.choice([0, 8, 256])),
                     chunk_queue_limit=int(rng.choice([0, 0, 3])))
            d1, st1 = ea.bfs(ctx, G, s, options=ea.Options(**o))
            d0, st0 = ea.bfs(ctx, G, s, options=ea.Options(call_every_edge=True, **o))
            assert (host(d1) == want).all() and (host(d0) == want).all(), (trial, n, m, s, o)
            assert st1.frontier_slots == st0.frontier_slots, (trial, n, m, s, o)
            assert st1.edges_traversed == st0.edges_traversed == int(deg[want != INF_I].sum())
