"""C++-level checks of include/gunrock/ that the C ABI does not reach (tests/cpp/engine_tests.hip):
frontier_t methods, parallel_for, enactor-overload swap rules, explicit-frontier advance for every
schedule, batch, rejection of unsupported variants -- and the same binary built with
-DGRX_ADVANCE_LB_OVERRIDE=bucketing; boundary_tests.hip covers the
reference include paths beside the hot path (launch_box, array, sample, smtx, print, memory)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.mark.parametrize("binary", ["engine_tests", "engine_tests_override", "boundary_tests"])
def test_cpp_engine_tests(binary):
    path = os.path.join(ROOT, "tests", "cpp", binary)
    if not os.path.exists(path):
        from essentials_amd.build import build_cpp_tests
        build_cpp_tests()
    r = subprocess.run([path], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "all passed" in r.stdout


def test_cpp_operator_results_against_the_oracle(tmp_path, oracle):
    """The device results of engine_tests' operator sections, dumped by the binary, checked HERE
    against oracle/ (the C restatement of the reference's operator semantics): parallel_for per
    vertex / per edge, explicit-frontier advance for every schedule (calls per destination and the
    emitted multiset), predicated filter, uniquify."""
    import json
    import numpy as np
    path = os.path.join(ROOT, "tests", "cpp", "engine_tests")
    if not os.path.exists(path):
        from essentials_amd.build import build_cpp_tests
        build_cpp_tests()
    out = tmp_path / "dump.json"
    r = subprocess.run([path, "--dump", str(out)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    d = json.load(open(out))
    Ap = np.array(d["row_offsets"], np.int32)
    Aj = np.array(d["column_indices"], np.int32)
    Ax = np.array(d["values"], np.float32)
    n = len(Ap) - 1
    # parallel_for: vertex -> degree, edge -> its source (orc_advance over the whole graph visits
    # every edge once and hands (src, dst, edge) to the callback)
    assert d["parallel_for_vertex_degrees"] == np.diff(Ap).tolist()
    src_of = np.full(len(Aj), -1, np.int64)

    def note(s, dst, e, w):
        src_of[e] = s
        return False
    oracle.advance(Ap, Aj, Ax, None, note, want_output=False)
    assert d["parallel_for_edge_sources"] == src_of.tolist()
    # explicit-frontier advance: op = count the call, keep (s + d) even
    frontier = np.array(d["advance_frontier"], np.int32)
    calls = np.zeros(n, np.int64)

    def op(s, dst, e, w):
        calls[dst] += 1
        return (s + dst) % 2 == 0
    want = oracle.advance(Ap, Aj, Ax, frontier, op)
    want = np.sort(want[want != -1]).tolist()
    schedules = [k[len("advance_output_"):] for k in d if k.startswith("advance_output_")]
    assert set(schedules) >= {"block_mapped", "merge_path", "bucketing", "thread_mapped", "warp_mapped",
                              "work_stealing"}
    for name in schedules:
        assert d["advance_output_" + name] == want, name
        assert d["advance_calls_per_destination_" + name] == calls.tolist(), name
    # settled hint (d % 3 == 0 by bitmap, d % 5 == 0 by predicate): the functor sees the other edges
    named = (np.arange(n) % 3 == 0) | (np.arange(n) % 5 == 0)
    calls[:] = 0

    def op_unnamed(s, dst, e, w):
        if named[dst]:
            return False
        calls[dst] += 1
        return (s + dst) % 2 == 0
    want = oracle.advance(Ap, Aj, Ax, frontier, op_unnamed)
    assert d["advance_settled_output"] == np.sort(want[want != -1]).tolist()
    assert d["advance_settled_calls_per_destination"] == calls.tolist()
    # neighborreduce: y[v] = sum over out-edges of w * x[dst], x[v] = v % 7 (exact in float)
    x = (np.arange(n) % 7).astype(np.float64)
    want_y = np.add.reduceat(np.concatenate([Ax * x[Aj], [0.0]]), np.minimum(Ap[:-1], len(Aj)))
    want_y[np.diff(Ap) == 0] = 0.0
    assert d["neighborreduce_sum_w_times_x"] == want_y.astype(np.int64).tolist()
    keep = oracle.filter_keep(d["filter_predicated_input"], lambda v: v == 8 or (v & 1))
    assert d["filter_predicated_output"] == keep.tolist()
    assert d["uniquify_output"] == oracle.uniquify(d["uniquify_input"]).tolist()
    # the product's rocPRIM sort call sites at 4 K - 16 K elements (the size range of the
    # round's GPU fault): frontier_t::sort, uniquify, transpose -- run once, checked here
    for n in (4096, 6000, 16384):
        src = np.array(d[f"sort_input_{n}"], np.int32)
        assert d[f"sort_output_{n}"] == np.sort(src, kind="stable").tolist(), n
        assert d[f"uniquify_output_{n}"] == oracle.uniquify(src).tolist(), n
    tAp = np.array(d["transpose_row_offsets"], np.int64)
    tAj = np.array(d["transpose_column_indices"], np.int64)
    assert 4096 <= len(tAj) <= 16384
    order = np.argsort(tAj, kind="stable")               # in-edges by destination, source order kept
    rows = np.repeat(np.arange(len(tAp) - 1), np.diff(tAp))
    assert d["transpose_result_edge_ids"] == order.tolist()
    assert d["transpose_result_indices"] == rows[order].tolist()
    assert d["transpose_result_offsets"] == np.concatenate(
        [[0], np.cumsum(np.bincount(tAj, minlength=len(tAp) - 1))]).tolist()
    # round-3 engine extensions, checked HERE against plain numpy restatements of their contracts
    n = len(Ap) - 1   # (the sort section above reused the name)
    assert d["select_range_output"] == [v for v in range(n) if v % 3 == 1]
    assert d["select_range_work_hint"] == [int(np.diff(Ap)[1::3].sum())]
    init = np.array(d["with_bounds_initial_labels"], np.int64)       # one min-relaxation round from the frontier
    want_lab = init.copy()
    for s_ in frontier[frontier >= 0]:
        for e in range(Ap[s_], Ap[s_ + 1]):
            want_lab[Aj[e]] = min(want_lab[Aj[e]], init[s_] + int(Ax[e]))
    # (sources hand on a read-only copy of their label: exactly one Jacobi round, order-independent)
    assert d["with_bounds_labels"] == want_lab.tolist()
    order = np.array(d["hot_first_vertex_of"], np.int64)
    assert sorted(order.tolist()) == list(range(n))
    deg = np.diff(Ap)
    assert (np.diff(deg[order]) <= 0).all()                          # falling degrees
    assert d["hot_first_offsets"] == np.concatenate([[0], np.cumsum(deg[order])]).tolist()
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    want_cols = np.concatenate([rank[Aj[Ap[v]:Ap[v + 1]]] for v in order]) if len(Aj) else np.zeros(0, np.int64)
    assert d["hot_first_indices"] == want_cols.tolist()
    # whole-graph advance walked by destination, after the weights were changed in place (every third
    # edge + 1): the sums per destination of w * (1 + source % 3), recomputed here from the arrays
    Ax2 = Ax.astype(np.float64).copy()
    Ax2[::3] += 1.0
    srcs = np.repeat(np.arange(n), np.diff(Ap))
    want_sums = np.bincount(Aj, weights=Ax2 * (1 + srcs % 3), minlength=n)
    assert d["by_destination_sums_reweighted"] == want_sums.astype(np.int64).tolist()
    assert d["failures"] == [0]
