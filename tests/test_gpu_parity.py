"""Parity of the HIP path (through the C ABI) against the oracle -- needs an MI355X.

Every test here calls libessentials_amd.so; the oracle (oracle/) is only the checker.
Integer / id results are compared bit-exactly; SSSP on integer-valued weights is
bit-exact too (the fix point is unique and every partial sum is exact); PageRank
uses an explicit tolerance (float atomics land in arbitrary order).
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_graph

# GRX_STRESS_SEED shifts the seeds of the randomised operator tests (soaks)
SEED_OFFSET = int(os.environ.get("GRX_STRESS_SEED", "0"))

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

INF_I = 2**31 - 1
INF_F = np.finfo(np.float32).max


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


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
    c = ea.Context(0)
    info = c.device_info()
    assert info["wavefront_size"] == 64
    return c


ALL_LB = ["block_mapped", "merge_path", "bucketing", "work_stealing", "thread_mapped", "warp_mapped"]
HOLES_LB = ["block_mapped", "merge_path", "thread_mapped"]


def host(t):
    return t.cpu().numpy()


# ---------------------------------------------------------------------------
# inputs: generator and loaders
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("scale,ef,seed,wseed,sym", [(8, 16, 1, 7, True), (10, 16, 1, 0, True),
                                                     (12, 16, 1, 7, True), (10, 8, 3, 5, False),
                                                     (14, 16, 1, 0, True)])
def test_rmat_generator_bit_exact(ea, ctx, oracle, scale, ef, seed, wseed, sym):
    g = ea.Graph.rmat(ctx, scale, ef, seed, wseed, sym)
    ap, aj, ax = g.to_host()
    n, Ap, Aj, Ax = oracle.rmat_csr(scale, ef, seed, wseed, sym)
    assert g.n_rows == n and g.nnz == len(Aj)
    assert (ap == Ap).all() and (aj == Aj).all() and (ax == Ax).all()


def test_mtx_loader_matches_oracle(ea, ctx, oracle, tmp_path):
    g = ea.Graph.from_mtx(os.path.join(GOLDEN_DIR, "chesapeake.mtx"))
    ap, aj, ax = g.to_host()
    n, Ap, Aj, Ax = oracle.mtx_to_csr(os.path.join(GOLDEN_DIR, "chesapeake.mtx"))
    assert (ap == Ap).all() and (aj == Aj).all() and (ax == Ax).all()
    p = str(tmp_path / "c.csr")
    g.write_csr_file(p)
    n2, m2, Ap2, Aj2, Ax2 = oracle.csr_read_binary(p)      # reference .csr layout
    assert n2 == 39 and (Ap2 == Ap).all() and (Aj2 == Aj).all() and (Ax2 == Ax).all()
    g2 = ea.Graph.from_csr_file(p)
    ap2, aj2, ax2 = g2.to_host()
    assert (ap2 == Ap).all() and (aj2 == Aj).all() and (ax2 == Ax).all()
    with pytest.raises(ea.EngineError):
        ea.Graph.from_mtx(str(tmp_path / "missing.mtx"))


# ---------------------------------------------------------------------------
# the three clients on the golden vectors
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("lb", ALL_LB)
def test_bfs_golden(ea, ctx, oracle, golden, lb):
    for name, g in golden.items():
        Ap, Aj, Ax = golden_graph(oracle, g)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        for run in g["runs"]:
            d, st = ea.bfs(ctx, G, run["source"], options=ea.Options(load_balance=ea.LoadBalance[lb]))
            d = host(d)
            assert sha(d) == run["bfs_sha256"], (name, run["source"], lb)
            assert st.vertices_reached == run["reached"]
            assert st.edges_traversed == run["edges_traversed"]
            # a push BFS expands every reached vertex exactly once
            assert st.edges_expanded == st.edges_traversed, (name, lb, st)
            # packed frontiers: one loop() per BFS level plus the empty-output one
            assert st.iterations == run["max_depth"] + 1, (name, lb, st)


@pytest.mark.parametrize("lb", HOLES_LB)
def test_bfs_golden_holes_layout(ea, ctx, oracle, golden, lb):
    """Reference output layout: one slot per traversed edge, -1 holes."""
    for name in ("chesapeake", "sample4x4", "tc4", "rmat8_w0", "rmat10_w7", "rmat10_directed"):
        g = golden[name]
        Ap, Aj, Ax = golden_graph(oracle, g)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        for run in g["runs"]:
            d, st = ea.bfs(ctx, G, run["source"],
                           options=ea.Options(load_balance=ea.LoadBalance[lb], holes_layout=True))
            assert sha(host(d)) == run["bfs_sha256"], (name, lb)
            _, tr = oracle.bfs_frontier(Ap, Aj, Ax, run["source"])
            assert st.iterations == tr.iterations
            assert st.frontier_slots == list(tr.frontier_slots[: tr.iterations]), (name, lb)


def test_chesapeake_reference_trace(ea, ctx):
    """4 loop() calls, input slots 1 / 11 / 124 / 205."""
    G = ea.Graph.from_mtx(os.path.join(GOLDEN_DIR, "chesapeake.mtx"))
    d, st = ea.bfs(ctx, G, 0, options=ea.Options(holes_layout=True))
    assert st.iterations == 4 and st.frontier_slots == [1, 11, 124, 205]
    assert host(d).tolist() == [0, 2, 2, 2, 2, 2, 1, 1, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1,
                                2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 2, 1, 2, 1]
    d, st = ea.bfs(ctx, G, 0)                       # packed (default) layout
    assert st.iterations == 3 and st.frontier_slots == [1, 11, 27]
    assert st.edges_traversed == 340 and st.vertices_reached == 39


@pytest.mark.parametrize("lb", ALL_LB)
def test_sssp_golden_bit_exact(ea, ctx, oracle, golden, lb):
    for name, g in golden.items():
        Ap, Aj, Ax = golden_graph(oracle, g)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        for run in g["runs"]:
            d, st = ea.sssp(ctx, G, run["source"], options=ea.Options(load_balance=ea.LoadBalance[lb]))
            assert sha(host(d).view(np.uint32)) == run["sssp_bits_sha256"], (name, run["source"], lb)
            # every reached vertex is expanded at least once; re-expanded when its distance improved
            assert st.edges_expanded >= st.edges_traversed == run["edges_traversed"], (name, lb, st)


def test_sssp_holes_layout(ea, ctx, oracle, golden):
    for name in ("chesapeake", "rmat8_w7", "rmat10_w7"):
        g = golden[name]
        Ap, Aj, Ax = golden_graph(oracle, g)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        for run in g["runs"]:
            d, st = ea.sssp(ctx, G, run["source"], options=ea.Options(holes_layout=True))
            assert sha(host(d).view(np.uint32)) == run["sssp_bits_sha256"], name


@pytest.mark.parametrize("lb", ["block_mapped", "merge_path", "bucketing"])
def test_pagerank_against_restatement(ea, ctx, oracle, lb):
    """Reference has no PageRank checker (parity unpinned); tolerance 5e-6 absolute on
    ranks that sum to 1 (float atomics land in arbitrary order)."""
    for scale, sym in ((10, False), (12, True)):
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, 8, 3, 0, sym)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        p, st = ea.pagerank(ctx, G, 0.85, 1e-6, options=ea.Options(load_balance=ea.LoadBalance[lb]))
        want, it = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
        p = host(p)
        assert abs(float(p.sum()) - 1.0) < 1e-3
        assert np.abs(p - want).max() < 5e-6, (scale, lb, np.abs(p - want).max())
        assert abs(st.iterations - it) <= 1


@pytest.mark.parametrize("lb", ["block_mapped", "merge_path"])
def test_pagerank_push_walked_by_destination(ea, oracle, monkeypatch, lb):
    """pr.hxx's push (a whole-graph advance without an output) walks the edges grouped by destination
    from its second iteration on, float adds of neighbouring lanes combined (operators/
    by_destination.hxx): same ranks as the restatement and as the row-by-row walk, on small graphs
    (forced), on one above the default threshold, and again from the remembered list."""
    for scale, ef, sym, force in ((10, 8, False, True), (12, 8, True, True), (17, 16, False, False)):
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, ef, 3, 5, sym)
        want, it = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
        results = {}
        for mode in ("0", "1" if force else None):
            if mode is None:
                monkeypatch.delenv("GRX_BY_DESTINATION", raising=False)   # the default: >= 2^20 edges
            else:
                monkeypatch.setenv("GRX_BY_DESTINATION", mode)
            c = ea.Context(0)                                             # reads the switch
            G = ea.Graph.from_host_csr(Ap, Aj, Ax)
            for again in range(2):                                        # second run: list remembered
                p, st = ea.pagerank(c, G, 0.85, 1e-6, options=ea.Options(load_balance=ea.LoadBalance[lb]))
                p = host(p)
                assert abs(float(p.sum()) - 1.0) < 1e-3
                assert np.abs(p - want).max() < 5e-6, (scale, lb, mode, again, np.abs(p - want).max())
                assert abs(st.iterations - it) <= 1
            results[mode] = p
            # the pull form walks the same sorted list from its second iteration on (one lookup per
            # edge); with the walk switched off it is the per-destination lists of round 2
            if not sym:
                G.build_in_edges(c)
            for again in range(2):
                q, sq = ea.pagerank(c, G, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
                assert sq.pull_iterations == sq.iterations
                assert np.abs(host(q) - want).max() < 5e-6, (scale, mode, again)
            G.close()
            c.close()
        a, b = results.values()
        assert np.abs(a - b).max() < 5e-6
        if not force:
            # at this size both forms ran on the hot-first copy of the graph (ranks handed over in the
            # caller's numbering); the caller's own numbering gives the same ranks
            monkeypatch.delenv("GRX_BY_DESTINATION", raising=False)
            monkeypatch.setenv("GRX_PR_HOT_FIRST", "0")
            c = ea.Context(0)
            G = ea.Graph.from_host_csr(Ap, Aj, Ax)
            for again in range(2):
                p, _ = ea.pagerank(c, G, 0.85, 1e-6, options=ea.Options(load_balance=ea.LoadBalance[lb]))
            assert np.abs(host(p) - want).max() < 5e-6 and np.abs(host(p) - b).max() < 5e-6
            G.build_in_edges(c)
            q, _ = ea.pagerank(c, G, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
            assert np.abs(host(q) - want).max() < 5e-6
            monkeypatch.setenv("GRX_PR_PULL_WALK", "0")      # round 2's per-destination lists
            q, _ = ea.pagerank(c, G, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
            assert np.abs(host(q) - want).max() < 5e-6
            monkeypatch.delenv("GRX_PR_HOT_FIRST")
            q, _ = ea.pagerank(c, G, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
            assert np.abs(host(q) - want).max() < 5e-6       # the lists need the caller's transpose
            monkeypatch.delenv("GRX_PR_PULL_WALK")
            G.close()
            c.close()
    monkeypatch.delenv("GRX_BY_DESTINATION", raising=False)


def test_pagerank_pull_matches_push_and_restatement(ea, ctx, oracle):
    """The pull form (sums per destination over in-edges, no atomics; new relative to pr.hxx)
    converges to the same ranks as the push scatter and the oracle's restatement; directed graphs
    get their transpose attached first (a graph without one is taken to be undirected)."""
    for scale, sym in ((10, False), (12, True), (13, False)):
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, 16, 3, 5, sym)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        if not sym:
            G.build_in_edges(ctx)
        p, st = ea.pagerank(ctx, G, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
        push, st_push = ea.pagerank(ctx, G, 0.85, 1e-6)
        want, it = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
        p, push = host(p), host(push)
        assert st.pull_iterations == st.iterations and st_push.pull_iterations == 0
        assert abs(float(p.sum()) - 1.0) < 1e-3
        assert np.abs(p - want).max() < 5e-6, (scale, np.abs(p - want).max())
        assert np.abs(p - push).max() < 5e-6
        assert abs(st.iterations - it) <= 1


# ---------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------
def _frontier_cases(n, rng):
    yield np.array([], np.int32)
    yield np.array([-1, -1, -1], np.int32)
    yield np.array([0], np.int32)
    yield np.arange(n, dtype=np.int32)
    f = rng.integers(0, n, 3 * n).astype(np.int32)       # duplicates are legal
    f[rng.random(len(f)) < 0.3] = -1                     # ragged with holes
    yield f
    yield rng.integers(0, n, 257).astype(np.int32)       # one element past a tile


@pytest.mark.parametrize("lb", ALL_LB)
@pytest.mark.parametrize("holes", [False, True])
def test_advance_matches_oracle(ea, ctx, torch, oracle, lb, holes):
    rng = np.random.default_rng(5 + SEED_OFFSET)
    for (scale, ef, sym) in ((6, 4, True), (9, 8, False), (11, 16, True)):
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, ef, 21, 9, sym)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        opts = ea.Options(load_balance=ea.LoadBalance[lb], holes_layout=holes, hub_threshold=64)
        for f in _frontier_cases(n, rng):
            calls = torch.zeros(max(len(Aj), 1), dtype=torch.int32, device="cuda")
            ft = torch.from_numpy(f).cuda()
            work = int(np.diff(Ap)[f[f != -1]].sum()) if len(f) else 0
            out = ea.advance(ctx, G, ft, ea.EdgeOp.count_edge, calls, 0, opts,
                             capacity=work + 16)
            want_calls = np.zeros(max(len(Aj), 1), np.int64)

            def op(s, d, e, w):
                want_calls[e] += 1
                return (s + d) % 3 == 0
            want = oracle.advance(Ap, Aj, Ax, f, op)
            got = host(out)
            assert (host(calls) == want_calls).all(), (lb, holes, scale, len(f))   # exactly once
            if holes:   # every schedule honours the layout (warp/bucketing route to merge_path)
                assert len(got) == len(want)
                assert sorted(got.tolist()) == sorted(want.tolist())
            else:
                assert sorted(got.tolist()) == sorted(want[want != -1].tolist()), (lb, holes, scale)
        # graph as the input frontier, no output (PageRank's form)
        acc = torch.zeros(n, dtype=torch.float32, device="cuda")
        ea.advance(ctx, G, None, ea.EdgeOp.sum_weight, acc, 0, opts, want_output=False)
        want_acc = np.zeros(n, np.float64)
        np.add.at(want_acc, Aj, Ax.astype(np.float64))
        assert np.array_equal(host(acc).astype(np.float64), want_acc)   # integer-valued weights: exact
        # graph as the input frontier with an output
        out = ea.advance(ctx, G, None, ea.EdgeOp.all, None, 0, opts)
        assert sorted(host(out).tolist()) == sorted(Aj.tolist())


def test_advance_hub_chunks(ea, ctx, torch, oracle):
    """A star: one list far above the hub threshold, cut into chunks."""
    n = 20000
    Ap = np.zeros(n + 1, np.int32); Ap[1:] = n - 1                # vertex 0 -> everyone else
    Aj = np.arange(1, n, dtype=np.int32); Ax = np.ones(n - 1, np.float32)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    for lb in ("block_mapped", "bucketing", "work_stealing"):
        calls = torch.zeros(n - 1, dtype=torch.int32, device="cuda")
        out = ea.advance(ctx, G, torch.tensor([0, -1, 0], dtype=torch.int32, device="cuda"),
                         ea.EdgeOp.count_edge, calls, 0, ea.Options(load_balance=ea.LoadBalance[lb]))
        assert (host(calls) == 2).all()
        want = [v for v in range(1, n) if v % 3 == 0] * 2
        assert sorted(host(out).tolist()) == sorted(want)


def test_advance_capacity_error(ea, ctx, torch, oracle):
    n, Ap, Aj, Ax = oracle.rmat_csr(8, 8, 2, 0, True)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    with pytest.raises(ea.EngineError):
        ea.advance(ctx, G, None, ea.EdgeOp.all, None, 0, capacity=8)


@pytest.mark.parametrize("alg", ["remove", "predicated", "compact", "bypass"])
def test_filter_matches_oracle(ea, ctx, torch, oracle, alg):
    rng = np.random.default_rng(3 + SEED_OFFSET)
    n, Ap, Aj, Ax = oracle.rmat_csr(8, 4, 1, 0, True)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    for size in (0, 1, 63, 64, 65, 1023, 1024, 1025, 5000, 100000):
        f = rng.integers(0, n, size).astype(np.int32)
        if size:
            f[rng.random(size) < 0.25] = -1
        ft = torch.from_numpy(f).cuda()
        calls = torch.zeros(n, dtype=torch.int32, device="cuda")
        got = host(ea.filter(ctx, G, ft, ea.FilterAlgorithm[alg], ea.VertexOp.count, calls))
        want_calls = np.bincount(f[f != -1], minlength=n)
        assert (host(calls) == want_calls).all()              # once per valid element, never on -1
        pred = lambda v: v % 3 != 0
        if alg == "bypass":
            assert got.tolist() == oracle.filter_bypass(f, pred).tolist()
        else:
            assert got.tolist() == oracle.filter_keep(f, pred).tolist()   # stable
    # the SSSP predicate (benign race: duplicates may survive, one copy at least is kept)
    f = rng.integers(0, n, 4000).astype(np.int32)
    stamp = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    got = host(ea.filter(ctx, G, torch.from_numpy(f).cuda(), ea.FilterAlgorithm.bypass,
                         ea.VertexOp.once, stamp, 7))
    kept = got[got != -1]
    assert set(kept.tolist()) == set(f.tolist()) and (host(stamp)[np.unique(f)] == 7).all()
    assert (got[got != -1] == f[got != -1]).all()


@pytest.mark.parametrize("alg", ["unique", "unique_copy"])
def test_uniquify_matches_oracle(ea, ctx, torch, oracle, alg):
    rng = np.random.default_rng(9 + SEED_OFFSET)
    for size in (0, 1, 2, 64, 1000, 1025, 70000):
        f = rng.integers(0, max(size // 3, 2), size).astype(np.int32)
        ft = torch.from_numpy(f).cuda()
        got = host(ea.uniquify(ctx, ft, ea.UniquifyAlgorithm[alg], best_effort=False))
        assert got.tolist() == oracle.uniquify(f, sort=True).tolist()
        got = host(ea.uniquify(ctx, ft, ea.UniquifyAlgorithm[alg], best_effort=True))
        assert got.tolist() == oracle.uniquify(f, sort=False).tolist()


# ---------------------------------------------------------------------------
# larger seeded graphs against the oracle (seconds of CPU)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("lb", ["block_mapped", "merge_path", "bucketing", "work_stealing"])
def test_bfs_sssp_rmat18_against_oracle(ea, ctx, oracle, lb):
    g = ea.Graph.rmat(ctx, 18, 16, seed=1, weight_seed=7)
    Ap, Aj, Ax = g.to_host()
    deg = np.diff(Ap)
    rng = np.random.default_rng(18 + SEED_OFFSET)
    sources = [0] + rng.choice(np.flatnonzero(deg > 0), 2).tolist()
    for s in sources:
        d, st = ea.bfs(ctx, g, int(s), options=ea.Options(load_balance=ea.LoadBalance[lb]))
        want, _ = oracle.bfs_heap(Ap, Aj, int(s))
        assert (host(d) == want).all(), (lb, s)
        assert st.edges_traversed == int(deg[want != INF_I].sum())
    w, st = ea.sssp(ctx, g, 0, options=ea.Options(load_balance=ea.LoadBalance[lb]))
    want, _ = oracle.sssp_heap(Ap, Aj, Ax, 0)
    assert (host(w).view(np.uint32) == want.view(np.uint32)).all(), lb


@pytest.mark.parametrize("scale,limit", [(16, 0), (20, 0), (20, 5)])
def test_bfs_settled_hint_changes_nothing(ea, ctx, oracle, scale, limit):
    """The search names its settled destinations on wide levels (operators/settled.hxx, the default)
    or calls the functor for every edge (call_every_edge): same depths, same counts, same frontier
    lengths level by level -- against the oracle and against each other, also with the hub chunk
    queue capped (hubs expanded in their tiles)."""
    g = ea.Graph.rmat(ctx, scale, 16, seed=3, weight_seed=0)
    Ap, Aj, _ = g.to_host()
    deg = np.diff(Ap)
    rng = np.random.default_rng(scale + SEED_OFFSET)
    for s in [0] + rng.choice(np.flatnonzero(deg > 0), 4).tolist():
        d1, st1 = ea.bfs(ctx, g, int(s), options=ea.Options(chunk_queue_limit=limit))
        d0, st0 = ea.bfs(ctx, g, int(s), options=ea.Options(call_every_edge=True, chunk_queue_limit=limit))
        want, _ = oracle.bfs_heap(Ap, Aj, int(s))
        assert (host(d1) == want).all() and (host(d0) == want).all(), s
        assert st1.frontier_slots == st0.frontier_slots, s
        assert st1.edges_traversed == st0.edges_traversed == int(deg[want != INF_I].sum())
        assert st1.vertices_reached == st0.vertices_reached == int((want != INF_I).sum())


def test_bfs_byte_labels_change_nothing(ea, ctx, oracle, monkeypatch):
    """Graphs beyond 2^22 vertices run the push search on one byte per vertex (clients.hxx); forced
    here onto small graphs: same depths, counts and frontier lengths as with the caller's 4-byte
    array, also past level 254, where the search continues in that array."""
    graphs = []
    g = ea.Graph.rmat(ctx, 17, 16, seed=5, weight_seed=0)
    graphs.append((g,) + tuple(g.to_host()[:2]))
    n = 1203                                     # a path with a few chords: 1000+ levels
    rows = np.arange(n - 1, dtype=np.int32)
    src = np.concatenate([rows, rows + 1, [0, 700]]).astype(np.int32)
    dst = np.concatenate([rows + 1, rows, [5, 1100]]).astype(np.int32)
    order = np.lexsort((dst, src))
    Ap = np.zeros(n + 1, np.int32)
    np.add.at(Ap, src + 1, 1)
    Ap = np.cumsum(Ap).astype(np.int32)
    Aj = np.ascontiguousarray(dst[order])
    graphs.append((ea.Graph.from_host_csr(Ap, Aj, np.ones(len(Aj), np.float32)), Ap, Aj))
    rng = np.random.default_rng(7 + SEED_OFFSET)
    for G, Ap, Aj in graphs:
        deg = np.diff(Ap)
        for s in [0] + rng.choice(np.flatnonzero(deg > 0), 3).tolist():
            want, _ = oracle.bfs_heap(Ap, np.ascontiguousarray(Aj), int(s))
            monkeypatch.setenv("GRX_BFS_BYTE_LABELS", "1")
            d1, st1 = ea.bfs(ctx, G, int(s))
            d1 = host(d1).copy()
            d2, _ = ea.bfs(ctx, G, int(s), options=ea.Options(call_every_edge=True,
                                                             load_balance=ea.LoadBalance.merge_path))
            d2 = host(d2).copy()
            monkeypatch.setenv("GRX_BFS_BYTE_LABELS", "0")
            d0, st0 = ea.bfs(ctx, G, int(s))
            assert (d1 == want).all() and (d2 == want).all() and (host(d0) == want).all(), s
            assert st1.frontier_slots == st0.frontier_slots and st1.iterations == st0.iterations
            assert st1.edges_traversed == st0.edges_traversed == int(deg[want != INF_I].sum())
        monkeypatch.delenv("GRX_BFS_BYTE_LABELS")


# ---------------------------------------------------------------------------
# BASELINE.json's full size (RMAT-22): size-independent properties
# ---------------------------------------------------------------------------
def test_bfs_rmat22_properties(ea, ctx, torch):
    g = ea.Graph.rmat(ctx, 22, 16, seed=1)
    assert g.n_rows == 1 << 22 and 130_000_000 < g.nnz < 2**27
    d0, st0 = ea.bfs(ctx, g, 0)
    # (0) with and without the settled-destination hint on the wide levels
    d, st = ea.bfs(ctx, g, 0, options=ea.Options(call_every_edge=True))
    assert torch.equal(d, d0) and st.frontier_slots == st0.frontier_slots
    # (1) every schedule produces the same labels
    for lb in ("merge_path", "bucketing", "work_stealing"):
        d, st = ea.bfs(ctx, g, 0, options=ea.Options(load_balance=ea.LoadBalance[lb]))
        assert torch.equal(d, d0), lb
        assert st.edges_traversed == st0.edges_traversed
    # (2) BFS certificate on the device: depth[src]=0; along every edge of a reached
    #     vertex the depth drops by at most one; every reached non-source vertex has a
    #     neighbour one level up (the graph is symmetric)
    h_ap, h_aj, _ = g.to_host()
    ap = torch.from_numpy(h_ap).cuda()
    aj = torch.from_numpy(np.ascontiguousarray(h_aj)).cuda()
    deg = (ap[1:] - ap[:-1]).long()
    src = torch.repeat_interleave(torch.arange(g.n_rows, device="cuda", dtype=torch.int32), deg)
    du = d0[src.long()].long(); dv = d0[aj.long()].long()
    reached_u = du != INF_I
    assert int(d0[0]) == 0
    assert bool(((dv[reached_u] != INF_I)).all())                     # closed under edges
    assert bool((dv[reached_u] <= du[reached_u] + 1).all())            # no edge skips a level
    best = torch.full((g.n_rows,), INF_I, dtype=torch.int64, device="cuda")
    best.scatter_reduce_(0, aj.long(), du, reduce="amin")              # min depth over in-neighbours
    reached = d0 != INF_I
    nonsrc = reached.clone(); nonsrc[0] = False
    assert bool((best[nonsrc] + 1 == d0[nonsrc].long()).all())         # tight: parent exists
    assert st0.vertices_reached == int(reached.sum())
    assert st0.edges_traversed == int(deg[reached].sum())


def test_sssp_rmat22_properties(ea, ctx, torch):
    """BASELINE configs[2] stand-in at full size (SSSP, bucketing, RMAT-22 with integer-valued
    weights in [1, 64]: every partial sum is exact, so the fix point is bit-exact and unique).
    (1) block_mapped, merge_path, bucketing and the reference's two-pass formulation give
    bit-identical distances; (2) the certificate of shortest-path distances, on the device:
    dist[src] = 0; dist[v] <= dist[u] + w along every edge of a reached u; every reached
    non-source vertex has a TIGHT in-edge (the graph is symmetric: in-edges = out-edges)."""
    g = ea.Graph.rmat(ctx, 22, 16, seed=1, weight_seed=7)
    src_vertex = 0
    w0, st0 = ea.sssp(ctx, g, src_vertex)
    for lb in ("merge_path", "bucketing"):
        w, st = ea.sssp(ctx, g, src_vertex, options=ea.Options(load_balance=ea.LoadBalance[lb]))
        assert torch.equal(w.view(torch.int32), w0.view(torch.int32)), lb
        assert st.edges_traversed == st0.edges_traversed
    w2, st2 = ea.sssp(ctx, g, src_vertex, options=ea.Options(sssp_two_pass=True))
    assert torch.equal(w2.view(torch.int32), w0.view(torch.int32))
    h_ap, h_aj, h_ax = g.to_host()
    ap = torch.from_numpy(h_ap).cuda()
    aj = torch.from_numpy(np.ascontiguousarray(h_aj)).cuda().long()
    ax = torch.from_numpy(np.ascontiguousarray(h_ax)).cuda()
    deg = (ap[1:] - ap[:-1]).long()
    u = torch.repeat_interleave(torch.arange(g.n_rows, device="cuda"), deg)
    INF = float(INF_F)
    du = w0[u]
    dv = w0[aj]
    reached_u = du < INF
    assert float(w0[src_vertex]) == 0.0
    assert bool((dv[reached_u] < INF).all())                            # closed under edges
    through = du[reached_u] + ax[reached_u]
    assert bool((dv[reached_u] <= through).all())                       # no edge can still relax
    best = torch.full((g.n_rows,), INF, dtype=torch.float32, device="cuda")
    cand = torch.where(reached_u, du + ax, torch.full_like(du, INF))
    best.scatter_reduce_(0, aj, cand, reduce="amin")                    # best candidate over in-edges
    reached = w0 < INF
    nonsrc = reached.clone()
    nonsrc[src_vertex] = False
    assert bool((best[nonsrc] == w0[nonsrc]).all())                     # tight: a parent exists
    assert st0.vertices_reached == int(reached.sum())
    assert st0.edges_traversed == int(deg[reached].sum())
    assert st0.edges_expanded >= st0.edges_traversed


# ---------------------------------------------------------------------------
# edge cases the reference's harness would hit (empty / degenerate inputs, bad arguments)
# ---------------------------------------------------------------------------
def test_degenerate_graphs(ea, ctx, torch):
    # vertices but no edges
    G = ea.Graph.from_host_csr(np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert G.n_rows == 5 and G.nnz == 0
    for lb in ALL_LB:
        d, st = ea.bfs(ctx, G, 3, options=ea.Options(load_balance=ea.LoadBalance[lb]))
        assert host(d).tolist() == [INF_I, INF_I, INF_I, 0, INF_I]
        assert st.iterations == 1 and st.edges_traversed == 0 and st.vertices_reached == 1
        w, _ = ea.sssp(ctx, G, 3, options=ea.Options(load_balance=ea.LoadBalance[lb]))
        assert host(w)[3] == 0 and host(w)[0] == INF_F
    p, st = ea.pagerank(ctx, G, 0.85, 1e-6)
    assert np.allclose(host(p), 0.2)                       # all dangling: uniform
    # a single self loop
    G = ea.Graph.from_host_csr(np.array([0, 1], np.int32), np.array([0], np.int32), np.ones(1, np.float32))
    d, st = ea.bfs(ctx, G, 0)
    assert host(d).tolist() == [0] and st.edges_traversed == 1
    # a path: one vertex per level, many supersteps
    n = 300
    ap = np.concatenate([np.arange(n, dtype=np.int32), [n - 1]]).astype(np.int32)
    aj = np.arange(1, n, dtype=np.int32)
    G = ea.Graph.from_host_csr(ap, aj, np.full(n - 1, 2.0, np.float32))
    d, st = ea.bfs(ctx, G, 0)
    assert host(d).tolist() == list(range(n)) and st.iterations == n
    w, _ = ea.sssp(ctx, G, 0, options=ea.Options(load_balance=ea.LoadBalance.merge_path))
    assert host(w).tolist() == [2.0 * i for i in range(n)]
    d, st = ea.bfs(ctx, G, 0, options=ea.Options(max_iterations=10))
    assert st.iterations == 10 and host(d)[10] == 10 and host(d)[11] == INF_I


def test_bad_arguments_are_errors(ea, ctx, torch, oracle):
    n, Ap, Aj, Ax = oracle.rmat_csr(6, 4, 1, 0)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    with pytest.raises(ea.EngineError):
        ea.bfs(ctx, G, -1)
    with pytest.raises(ea.EngineError):
        ea.sssp(ctx, G, n)
    with pytest.raises(ea.EngineError):
        ea.advance(ctx, G, None, ea.EdgeOp.bfs, None)       # functor needs its state
    f = torch.zeros(4, dtype=torch.int32, device="cuda")
    with pytest.raises(ea.EngineError):
        ea.advance(ctx, G, f, ea.EdgeOp.all, options=ea.Options(load_balance=17))
    with pytest.raises(ea.EngineError):
        ea.Context(99)


def test_independent_contexts_do_not_interfere(ea, torch, oracle):
    """Two contexts (two streams) interleaved on one thread and two host threads at once
    (the reference's operators::batch pattern)."""
    import threading
    n, Ap, Aj, Ax = oracle.rmat_csr(14, 16, 1, 7)
    want, _ = oracle.bfs_heap(Ap, np.ascontiguousarray(Aj), 0)
    results = {}

    def work(k):
        c = ea.Context(0)
        g = ea.Graph.from_host_csr(Ap, Aj, Ax)
        for _ in range(5):
            d, _ = ea.bfs(c, g, 0)
            results[k] = bool((host(d) == want).all())
            if not results[k]:
                return
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert results == {0: True, 1: True, 2: True, 3: True}


# ---------------------------------------------------------------------------
# pull advance + direction-optimising BFS (new relative to the reference)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("lb", ["block_mapped", "merge_path"])
def test_direction_optimized_bfs_matches_oracle(ea, ctx, oracle, golden, lb):
    opts = ea.Options(load_balance=ea.LoadBalance[lb], direction_optimized=True)
    for name, g in golden.items():
        Ap, Aj, Ax = golden_graph(oracle, g)
        G = ea.Graph.from_host_csr(Ap, Aj, Ax)
        if name in ("rmat10_directed", "sample4x4"):
            G.build_in_edges(ctx)         # directed: pull walks the attached transpose
        for run in g["runs"]:
            d, st = ea.bfs(ctx, G, run["source"], options=opts)
            assert sha(host(d)) == run["bfs_sha256"], (name, run["source"])
            assert st.edges_traversed == run["edges_traversed"]
    # forced pulling from the first possible level on, and never
    n, Ap, Aj, Ax = oracle.rmat_csr(16, 16, 1, 0)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    want, _ = oracle.bfs_heap(Ap, np.ascontiguousarray(Aj), 5)
    # a directed graph, pulling forced: in-edges come from graph::build::transpose
    nd, Apd, Ajd, Axd = oracle.rmat_csr(13, 8, 5, 0, False)
    Gd = ea.Graph.from_host_csr(Apd, Ajd, Axd).build_in_edges(ctx)
    for s in (0, 1, 77):
        wantd, _ = oracle.bfs_heap(Apd, np.ascontiguousarray(Ajd), s)
        d, st = ea.bfs(ctx, Gd, s, options=ea.Options(load_balance=ea.LoadBalance[lb],
                                                      direction_optimized=True, do_alpha=1e9, do_beta=1e9))
        assert (host(d) == wantd).all(), ("directed", s)
    for alpha, beta in ((1e9, 1e9), (1e-9, 24.0), (14.0, 24.0), (2.0, 2.0)):
        d, st = ea.bfs(ctx, G, 5, options=ea.Options(load_balance=ea.LoadBalance[lb],
                                                      direction_optimized=True, do_alpha=alpha, do_beta=beta))
        assert (host(d) == want).all(), (alpha, beta)
        if alpha == 1e9:
            assert st.pull_iterations >= 2
        if alpha == 1e-9:
            assert st.pull_iterations == 0


def test_direction_optimized_bfs_rmat22(ea, ctx, torch):
    g = ea.Graph.rmat(ctx, 22, 16, seed=1)
    d0, st0 = ea.bfs(ctx, g, 0)
    d1, st1 = ea.bfs(ctx, g, 0, options=ea.Options(direction_optimized=True))
    assert torch.equal(d0, d1) and st1.pull_iterations >= 1
    assert st1.edges_traversed == st0.edges_traversed


def test_sssp_fractional_weights_within_one_ulp(ea, ctx, torch, oracle):
    """north_star: SSSP distances within 1 ULP.  With arbitrary float weights every candidate
    distance is a left-to-right path sum and float addition is monotone, so the fix point is the
    same as the reference checker's; the test allows 1 ULP and reports exact equality."""
    rng = np.random.default_rng(4 + SEED_OFFSET)
    n, Ap, Aj, _ = oracle.rmat_csr(14, 16, 1, 0)
    Aj = np.ascontiguousarray(Aj)
    Ax = (rng.random(len(Aj)) * 9.9 + 0.1).astype(np.float32)
    G = ea.Graph.from_host_csr(Ap, Aj, Ax)
    for lb in ("block_mapped", "merge_path", "bucketing"):
        w, _ = ea.sssp(ctx, G, 7217, options=ea.Options(load_balance=ea.LoadBalance[lb]))
        want, _ = oracle.sssp_heap(Ap, Aj, Ax, 7217)
        got = host(w)
        finite = want < 3e38
        assert ((got < 3e38) == finite).all()
        ulp = np.abs(got[finite].view(np.int32).astype(np.int64) - want[finite].view(np.int32).astype(np.int64))
        assert ulp.max() <= 1, (lb, int(ulp.max()))


# ---------------------------------------------------------------------------
# hot-first numbering (include/gunrock/graph/reorder.hxx): labels come back in the CALLER's ids
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("scale,seed", [(10, 1), (14, 3)])
def test_hot_first_numbering_changes_nothing(ea, ctx, oracle, monkeypatch, scale, seed):
    """grx_graph_hot_first(1): BFS (4-byte, byte and direction-optimised labels) and SSSP (packed
    and two-word labels) on the renumbered copy deliver exactly the oracle's labels per CALLER id;
    the forms that stand for the unchanged reference clients keep the caller's graph."""
    g = ea.Graph.rmat(ctx, scale, 16, seed, 7)
    Ap, Aj, Ax = g.to_host()
    deg = np.diff(Ap)
    rng = np.random.default_rng(11 + SEED_OFFSET)
    sources = [0] + [int(s) for s in rng.choice(np.flatnonzero(deg > 0), 3, replace=False)]
    g.hot_first(ctx, True)
    for s in sources:
        want_d, _ = oracle.bfs_heap(Ap, Aj, s)
        want_w, _ = oracle.sssp_heap(Ap, Aj, Ax, s)
        for env in ({}, {"GRX_BFS_BYTE_LABELS": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            d, st = ea.bfs(ctx, g, s)
            assert (host(d) == want_d).all(), (s, env)
            assert st.edges_traversed == int(deg[want_d != INF_I].sum())
            for k in env:
                monkeypatch.delenv(k)
        d, _ = ea.bfs(ctx, g, s, options=ea.Options(direction_optimized=True))
        assert (host(d) == want_d).all(), (s, "direction optimised")
        for lb in ("block_mapped", "merge_path", "bucketing"):
            d, _ = ea.bfs(ctx, g, s, options=ea.Options(load_balance=ea.LoadBalance[lb]))
            assert (host(d) == want_d).all(), (s, lb)
        for packed in ("1", "0"):
            monkeypatch.setenv("GRX_SSSP_PACKED", packed)
            w, _ = ea.sssp(ctx, g, s)
            assert (host(w).view(np.uint32) == want_w.view(np.uint32)).all(), (s, packed)
        monkeypatch.delenv("GRX_SSSP_PACKED")
        w, _ = ea.sssp(ctx, g, s, options=ea.Options(sssp_two_pass=True))
        assert (host(w).view(np.uint32) == want_w.view(np.uint32)).all()
        d, _ = ea.bfs(ctx, g, s, options=ea.Options(call_every_edge=True))
        assert (host(d) == want_d).all()
    # dropping the copy changes nothing either
    g.hot_first(ctx, False)
    d, _ = ea.bfs(ctx, g, sources[1])
    assert (host(d) == oracle.bfs_heap(Ap, Aj, sources[1])[0]).all()


def test_hot_first_renumbering_is_a_degree_sorted_isomorphism(ea, ctx, torch):
    """The renumbered copy itself (C++ surface: graph::build::hot_first through the handle): BFS
    depths of the copy's own numbering are a permutation of the caller's, and the wide-level forms
    (settled bitmap, SSSP bound image) that read the low ids see the heaviest vertices there."""
    g = ea.Graph.rmat(ctx, 16, 16, 1, 7)          # 65,536 vertices, ~2 M edges: the automatic rule applies
    Ap = g.offsets_to_host()
    d0, st0 = ea.bfs(ctx, g, 0)                   # automatic: runs on the copy
    g.hot_first(ctx, False)
    d1, st1 = ea.bfs(ctx, g, 0)                   # caller's numbering
    assert torch.equal(d0, d1) and st0.edges_traversed == st1.edges_traversed
    assert st0.frontier_slots == st1.frontier_slots
    w0, _ = ea.sssp(ctx, g, 5)
    g.hot_first(ctx, True)
    w1, _ = ea.sssp(ctx, g, 5)
    assert torch.equal(w0.view(torch.int32), w1.view(torch.int32))
