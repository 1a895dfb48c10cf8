"""The oracle against the reference's own known answers.

Pins oracle/grx_oracle.c to: (i) the golden vectors produced by the reference's
bfs_cpu.hxx / sssp_cpu.hxx (tests/golden/golden.json), (ii) the chesapeake
numbers and frontier trace, (iii) when oracle/_ref is present, the reference
checkers run live on fresh seeded graphs.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_graph

INF_I = 2**31 - 1


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_chesapeake_csr_known_answers(oracle):
    n, Ap, Aj, Ax = oracle.mtx_to_csr(os.path.join(GOLDEN_DIR, "chesapeake.mtx"))
    assert n == 39 and len(Aj) == 340
    assert Ap[:6].tolist() == [0, 11, 22, 29, 33, 37]
    assert Aj[:11].tolist() == [6, 7, 10, 11, 12, 21, 22, 33, 34, 36, 38]
    assert (Ax == 1.0).all()  # pattern file
    deg = np.diff(Ap)
    assert deg.min() == 3 and deg.max() == 33


def test_chesapeake_bfs_depths(oracle):
    _, Ap, Aj, Ax = oracle.mtx_to_csr(os.path.join(GOLDEN_DIR, "chesapeake.mtx"))
    want = [0, 2, 2, 2, 2, 2, 1, 1, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 2, 2, 2, 2, 2, 2, 2,
            2, 2, 2, 1, 1, 2, 1, 2, 1]
    d, _ = oracle.bfs_heap(Ap, Aj, 0)
    assert d.tolist() == want
    w, _ = oracle.sssp_heap(Ap, Aj, Ax, 0)
    assert w.tolist() == [float(x) for x in want]


def test_chesapeake_frontier_trace(oracle):
    """Reference-semantics (one slot per traversed edge) trace, SURVEY.md section 4."""
    _, Ap, Aj, Ax = oracle.mtx_to_csr(os.path.join(GOLDEN_DIR, "chesapeake.mtx"))
    d, t = oracle.bfs_frontier(Ap, Aj, Ax, 0)
    assert t.iterations == 4 and t.edges_traversed == 340
    assert list(t.frontier_slots[:4]) == [1, 11, 124, 205]
    assert list(t.frontier_valid[:4]) == [1, 11, 27, 0]


def test_golden_all_graphs(oracle, golden):
    for name, g in golden.items():
        Ap, Aj, Ax = golden_graph(oracle, g)
        assert sha(Ap) == g["row_offsets_sha256"], name
        assert sha(Aj) == g["col_sha256"], name
        assert sha(Ax) == g["val_sha256"], name
        for run in g["runs"]:
            s = run["source"]
            d, _ = oracle.bfs_heap(Ap, Aj, s)
            assert sha(d) == run["bfs_sha256"], (name, s)
            df, tr = oracle.bfs_frontier(Ap, Aj, Ax, s)
            assert (df == d).all(), (name, s)
            assert tr.edges_traversed == run["edges_traversed"], (name, s)
            assert tr.iterations == run["max_depth"] + 2 or run["reached"] == 1 or \
                tr.iterations == run["max_depth"] + 1, (name, s)
            w, _ = oracle.sssp_heap(Ap, Aj, Ax, s)
            assert sha(w.view(np.uint32)) == run["sssp_bits_sha256"], (name, s)
            wf, _ = oracle.sssp_frontier(Ap, Aj, Ax, s)
            assert (wf.view(np.uint32) == w.view(np.uint32)).all(), (name, s)
            if "bfs" in run:
                assert d.tolist() == run["bfs"]
            reached = d[d != INF_I]
            assert len(reached) == run["reached"]
            assert np.bincount(reached).tolist() == run["depth_hist"]


def test_live_against_reference_checkers(oracle):
    from oracle.oracle import RefOracle
    if not RefOracle.available():
        pytest.skip("oracle/_ref not built (reference tree not mounted)")
    ref = RefOracle()
    for scale, seed, wseed in ((9, 11, 3), (11, 12, 0), (13, 13, 9)):
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, 8, seed, wseed)
        rng = np.random.default_rng(seed)
        for s in rng.integers(0, n, 3):
            d, _ = oracle.bfs_heap(Ap, Aj, int(s)); dr, _ = ref.bfs(Ap, Aj, int(s))
            assert (d == dr).all()
            w, _ = oracle.sssp_heap(Ap, Aj, Ax, int(s)); wr, _ = ref.sssp(Ap, Aj, Ax, int(s))
            assert (w.view(np.uint32) == wr.view(np.uint32)).all()


def test_levelsync_baseline_matches(oracle):
    n, Ap, Aj, Ax = oracle.rmat_csr(12, 16, 5, 0)
    d, _ = oracle.bfs_heap(Ap, Aj, 3)
    dl, _, th = oracle.bfs_levelsync(Ap, Aj, 3)
    assert (d == dl).all() and th >= 1


def test_rmat_generator_spec(oracle):
    # integer-exact spec: pair k of (scale, seed) is a pure function
    assert oracle.rmat_pair(10, 1, 0) == oracle.rmat_pair(10, 1, 0)
    n, Ap, Aj, Ax = oracle.rmat_csr(8, 16, seed=1, weight_seed=7)
    pairs = [oracle.rmat_pair(8, 1, k) for k in range(16 << 8)]
    # loader-style symmetrisation: (u,v),(v,u); self loop once; duplicates kept
    want = sum(2 if u != v else 1 for u, v in pairs)
    assert len(Aj) == want == Ap[-1]
    rows = {}
    for k, (u, v) in enumerate(pairs):
        w = float(oracle.L.orc_rmat_weight(7, k))
        rows.setdefault(u, []).append((v, w))
        if u != v:
            rows.setdefault(v, []).append((u, w))
    for r, lst in rows.items():
        assert Aj[Ap[r]:Ap[r + 1]].tolist() == [c for c, _ in lst]
        assert Ax[Ap[r]:Ap[r + 1]].tolist() == [w for _, w in lst]
    assert Ax.min() >= 1 and Ax.max() <= 64 and (Ax == np.round(Ax)).all()


def test_csr_binary_roundtrip(oracle, tmp_path):
    n, Ap, Aj, Ax = oracle.rmat_csr(8, 4, 2, 3)
    p = str(tmp_path / "g.csr")
    oracle.csr_write_binary(p, n, n, Ap, np.ascontiguousarray(Aj), np.ascontiguousarray(Ax))
    raw = np.fromfile(p, np.int32)
    assert raw[:3].tolist() == [n, n, len(Aj)]          # header of formats/csr.hxx:202-210
    assert os.path.getsize(p) == 12 + 4 * (n + 1) + 8 * len(Aj)
    n2, m2, Ap2, Aj2, Ax2 = oracle.csr_read_binary(p)
    assert n2 == n and (Ap2 == Ap).all() and (Aj2 == Aj).all() and (Ax2 == Ax).all()


def test_mtx_loader_variants(oracle, tmp_path):
    p = tmp_path / "a.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% c\n3 3 3\n1 2 0.5\n3 1 2\n2 2 7\n")
    n, m, r, c, v = oracle.mtx_load(str(p))
    assert (n, m) == (3, 3) and r.tolist() == [0, 2, 1] and c.tolist() == [1, 0, 1]
    assert v.tolist() == [0.5, 2.0, 7.0]
    p.write_text("%%MatrixMarket matrix coordinate integer symmetric\n3 3 3\n2 1 4\n3 3 9\n3 1 5\n")
    n, m, r, c, v = oracle.mtx_load(str(p))
    # off-diagonals doubled in place, diagonal once (io/matrix_market.hxx:213-230)
    assert r.tolist() == [1, 0, 2, 2, 0] and c.tolist() == [0, 1, 2, 0, 2]
    assert v.tolist() == [4, 4, 9, 5, 5]
    p.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(RuntimeError):
        oracle.mtx_load(str(p))


def test_operator_restatements(oracle):
    Ap = np.array([0, 0, 2, 3, 4], np.int32); Aj = np.array([0, 1, 2, 1], np.int32)
    Ax = np.array([5, 8, 3, 6], np.float32)
    calls = []
    out = oracle.advance(Ap, Aj, Ax, [1, -1, 3, 0], lambda s, d, e, w: (calls.append((s, d, e, w)), d != 1)[1])
    assert calls == [(1, 0, 0, 5.0), (1, 1, 1, 8.0), (3, 1, 3, 6.0)]   # invalid and degree-0 slots: no calls
    assert out.tolist() == [0, -1, -1]
    out = oracle.advance(Ap, Aj, Ax, None, lambda s, d, e, w: True)    # graph as frontier
    assert out.tolist() == [0, 1, 2, 1]
    seen = []
    assert oracle.filter_bypass([3, -1, 2, 3], lambda v: (seen.append(v), v != 2)[1]).tolist() == [3, -1, -1, 3]
    assert seen == [3, 2, 3]                                           # predicate never sees invalids
    assert oracle.filter_keep([3, -1, 2, 3, 0], lambda v: v != 2).tolist() == [3, 3, 0]
    assert oracle.uniquify([5, 1, 5, 5, 1, -1], sort=True).tolist() == [-1, 1, 5]
    assert oracle.uniquify([5, 1, 5, 5, 1], sort=False).tolist() == [5, 1, 5, 1]


def test_pagerank_restatement(oracle):
    """Parity UNPINNED in the reference (no checker, no test); sanity only."""
    n, Ap, Aj, Ax = oracle.rmat_csr(10, 8, 3, 0, symmetrize=False)
    p, it = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
    assert it > 1 and abs(float(p.sum()) - 1.0) < 1e-3 and (p > 0).all()
    # power-iteration fixed point in float64
    deg = np.diff(Ap).astype(np.float64)
    q = np.full(n, 1.0 / n)
    src = np.repeat(np.arange(n), np.diff(Ap))
    for _ in range(200):
        contrib = np.where(deg > 0, 0.85 * q / np.maximum(deg, 1), 0)
        nq = np.full(n, (1 - 0.85 + 0.85 * q[deg == 0].sum()) / n)
        np.add.at(nq, Aj, contrib[src])
        q = nq
    assert np.abs(p - q).max() < 5e-5
