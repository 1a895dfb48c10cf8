"""Host-side contracts of the C ABI / Python layer that the kernels' parity tests do not reach:
stream ordering between torch and the engine's private stream, pull traversals refused on directed
graphs without in-edges, the device block cache."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

INF_I = 2**31 - 1


@pytest.fixture(scope="module")
def env():
    import torch
    import essentials_amd as ea
    assert torch.cuda.is_available()
    return ea, torch, ea.Context(0)


def _busy(torch, ms_target=30):
    """Keep torch's current stream busy for a while (big matmuls), so that anything enqueued behind
    it has certainly NOT run when the next host call is made."""
    a = torch.randn(8192, 8192, device="cuda")
    for _ in range(max(ms_target // 3, 1)):
        a = (a @ a).clamp_(-1, 1)
    return a


def test_engine_waits_for_operands_torch_is_still_writing(env, oracle):
    """The default Context runs on a private NON-BLOCKING stream.  Operands that torch is still
    filling on its own stream must be complete before an engine kernel reads them: the wrappers
    order the engine's stream after torch's (grx_context_wait_stream), on the device."""
    ea, torch, ctx = env
    n, Ap, Aj, Ax = oracle.rmat_csr(14, 8, 5, 0, True)
    g = ea.Graph.from_host_csr(Ap, Aj, Ax)
    frontier = torch.arange(0, n, 3, dtype=torch.int32, device="cuda")
    for trial in range(3):
        depth = torch.empty(n, dtype=torch.int32, device="cuda")
        depth.fill_(0)              # wrong on purpose: every vertex "visited"
        torch.cuda.synchronize()
        keep = _busy(torch)
        depth.fill_(INF_I)          # queued behind ~30 ms of matmuls on torch's stream
        depth[frontier.long()] = 0
        out = ea.advance(ctx, g, frontier, ea.EdgeOp.bfs, state=depth, iparam=0)
        torch.cuda.synchronize()
        del keep
        d = depth.cpu().numpy()
        want = np.full(n, INF_I, np.int64)
        f = frontier.cpu().numpy()
        want[f] = 0
        for v in f:
            nb = Aj[Ap[v]:Ap[v + 1]]
            want[nb] = np.minimum(want[nb], 1)
        assert (d == want).all(), f"trial {trial}: the advance read labels torch had not written yet"
        assert set(out.cpu().numpy().tolist()) == set(np.flatnonzero(want == 1).tolist())
    # uniquify clones its input on torch's stream, then sorts on the engine's
    x = torch.randint(0, 1000, (200000,), dtype=torch.int32, device="cuda")
    keep = _busy(torch)
    y = x * 1                       # behind the matmuls
    u = ea.uniquify(ctx, y)
    torch.cuda.synchronize()
    assert np.array_equal(np.sort(u.cpu().numpy()), np.unique(x.cpu().numpy()))
    g.close()


def test_context_on_torch_stream_needs_no_ordering_call(env, oracle):
    ea, torch, _ = env
    s = torch.cuda.Stream()
    ctx = ea.Context(0, stream=s.cuda_stream)
    n, Ap, Aj, Ax = oracle.rmat_csr(12, 8, 3, 7, True)
    g = ea.Graph.from_host_csr(Ap, Aj, Ax)
    with torch.cuda.stream(s):
        d, _ = ea.bfs(ctx, g, 0)
    want, _ = oracle.bfs_heap(Ap, Aj, 0)
    assert (d.cpu().numpy() == want).all()
    g.close()


def test_pull_is_refused_on_a_directed_graph_without_in_edges(env, oracle):
    """A CSR of unknown symmetry is verified on the device the first time a pull traversal asks;
    a directed one is refused (GRX_ERR_UNSUPPORTED = -3) instead of silently walking out-edges as
    in-edges; with in-edges attached, or when the CSR is its own transpose, it runs."""
    ea, torch, ctx = env
    n, Ap, Aj, Ax = oracle.rmat_csr(11, 8, 3, 0, False)       # directed
    g = ea.Graph.from_host_csr(Ap, Aj, Ax)
    do = ea.Options(direction_optimized=True)
    with pytest.raises(ea.EngineError, match=r"\(-3\).*in-edges"):
        ea.bfs(ctx, g, 0, options=do)
    with pytest.raises(ea.EngineError, match=r"\(-3\)"):
        ea.pagerank(ctx, g, options=do)
    d_push, _ = ea.bfs(ctx, g, 0)                              # push is unaffected
    g.build_in_edges(ctx)
    d_pull, st = ea.bfs(ctx, g, 0, options=do)
    assert torch.equal(d_push, d_pull)
    want, _ = oracle.bfs_heap(Ap, Aj, 0)
    assert (d_pull.cpu().numpy() == want).all()
    g.close()
    # the generator marks its directed graphs without any check
    gd = ea.Graph.rmat(ctx, 10, 8, 3, 0, False)
    with pytest.raises(ea.EngineError, match=r"\(-3\)"):
        ea.bfs(ctx, gd, 0, options=do)
    gd.close()
    # a symmetric CSR handed over as plain arrays passes the device check
    n, Ap, Aj, Ax = oracle.rmat_csr(11, 8, 3, 0, True)
    gs = ea.Graph.from_host_csr(Ap, Aj, Ax)
    d, st = ea.bfs(ctx, gs, 0, options=do)
    want, _ = oracle.bfs_heap(Ap, Aj, 0)
    assert (d.cpu().numpy() == want).all()
    # one asymmetric edge more: refused
    Aj2 = np.concatenate([Aj, [5]]).astype(np.int32)
    Ax2 = np.concatenate([Ax, [1.0]]).astype(np.float32)
    Ap2 = Ap.copy()
    Ap2[-1] += 1            # the extra edge (n-1 -> 5) joins the last row
    if 5 in Aj[Ap[n - 1]:Ap[n]] or (n - 1) in Aj[Ap[5]:Ap[6]]:
        pytest.skip("edge already present")
    ga = ea.Graph.from_host_csr(Ap2, Aj2, Ax2)
    with pytest.raises(ea.EngineError, match=r"\(-3\)"):
        ea.bfs(ctx, ga, 0, options=do)
    gs.close()
    ga.close()


def test_block_cache_gives_memory_back(env, oracle):
    """Frontier blocks of finished runs are parked for reuse; grx_trim_cache returns them, and a
    graph's own arrays are never parked."""
    ea, torch, ctx = env
    ea.Context.trim_cache()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    g = ea.Graph.rmat(ctx, 20, 16, 1, 7)
    ea.bfs(ctx, g, 0)                   # two 1.5 * |E| frontiers: ~200 MB parked afterwards
    free_run, _ = torch.cuda.mem_get_info()
    g.close()
    free_closed, _ = torch.cuda.mem_get_info()
    assert free_closed - free_run > 200e6, "closing a graph must free its CSR arrays, not park them"
    ea.Context.trim_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 - free_closed > 150e6, "trim must return the parked frontier blocks"
    assert abs(free1 - free0) < 64e6
