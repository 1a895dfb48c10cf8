"""The reference's own bfs.hxx / sssp.hxx / pr.hxx, UNCHANGED, running on this engine.

oracle/_ref/libgrx_ref_clients.so is compiled in the build container from the reference headers
where they lie (oracle/ref_build.sh) with this repository's include/ first on the include path;
it is a built artefact (never the reference's source) and is skipped when absent."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_graph

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def refc():
    from oracle.oracle import RefClients
    if not RefClients.available():
        pytest.skip("oracle/_ref/libgrx_ref_clients.so not built (reference tree was not mounted)")
    import torch
    assert torch.cuda.is_available()
    return RefClients()


def dev(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def test_reference_bfs_sssp_headers_on_golden(refc, oracle, golden):
    import torch
    for name in ("chesapeake", "sample4x4", "tc4", "rmat8_w7", "rmat10_w7", "rmat12_w7", "rmat14_w0",
                 "rmat10_directed"):
        g = golden[name]
        Ap, Aj, Ax = golden_graph(oracle, g)
        ap, aj, ax = dev(Ap), dev(Aj), dev(Ax)
        for run in g["runs"]:
            d = torch.empty(len(Ap) - 1, dtype=torch.int32, device="cuda")
            refc.bfs(ap, aj, ax, run["source"], d)
            assert sha(d.cpu().numpy()) == run["bfs_sha256"], (name, run["source"])
            w = torch.empty(len(Ap) - 1, dtype=torch.float32, device="cuda")
            refc.sssp(ap, aj, ax, run["source"], w)
            assert sha(w.cpu().numpy().view(np.uint32)) == run["sssp_bits_sha256"], (name, run["source"])


def test_reference_pr_header(refc, oracle):
    import torch
    n, Ap, Aj, Ax = oracle.rmat_csr(11, 8, 3, 0, False)
    p = torch.empty(n, dtype=torch.float32, device="cuda")
    refc.pr(dev(Ap), dev(Aj), dev(Ax), 0.85, 1e-6, p)
    want, _ = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
    assert np.abs(p.cpu().numpy() - want).max() < 5e-6   # parity unpinned in the reference


def test_reference_pr_header_walked_by_destination(refc, oracle, monkeypatch):
    """The unchanged pr.hxx with its whole-graph advance walked by destination from the second
    iteration on (forced on the small graph; the default from 2^20 edges): same ranks."""
    import torch
    for scale, ef, force in ((11, 8, True), (17, 16, False)):
        if force:
            monkeypatch.setenv("GRX_BY_DESTINATION", "1")
        else:
            monkeypatch.delenv("GRX_BY_DESTINATION", raising=False)
        n, Ap, Aj, Ax = oracle.rmat_csr(scale, ef, 3, 0, False)
        p = torch.empty(n, dtype=torch.float32, device="cuda")
        refc.pr(dev(Ap), dev(Aj), dev(Ax), 0.85, 1e-6, p)
        want, _ = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
        assert np.abs(p.cpu().numpy() - want).max() < 5e-6, scale
    monkeypatch.delenv("GRX_BY_DESTINATION", raising=False)


def test_reference_bfs_header_rmat20(refc, oracle):
    """Same engine, reference client: a graph big enough for hubs, chunks and several levels."""
    import torch
    import essentials_amd as ea
    ctx = ea.Context(0)
    g = ea.Graph.rmat(ctx, 20, 16, 1, 7)
    Ap, Aj, Ax = g.to_host()
    ap, aj, ax = dev(Ap), dev(Aj), dev(Ax)
    d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
    ms = refc.bfs(ap, aj, ax, 0, d)
    want, cpu_ms = oracle.bfs_heap(Ap, Aj, 0)
    assert (d.cpu().numpy() == want).all()
    mine, st = ea.bfs(ctx, g, 0)
    assert torch.equal(mine, d)
    print(f"reference bfs.hxx on this engine: {ms:.3f} ms; conformance client {st.elapsed_ms:.3f} ms; "
          f"reference CPU checker port {cpu_ms:.0f} ms")


def test_reference_sssp_header_with_bucketing_override(oracle, golden):
    """BASELINE config 3: sssp.hxx unchanged + -DGRX_ADVANCE_LB_OVERRIDE=bucketing."""
    import torch
    from oracle.oracle import RefClients
    if not RefClients.available("bucketing"):
        pytest.skip("oracle/_ref/libgrx_ref_clients_bucketing.so not built")
    rc = RefClients("bucketing")
    for name in ("chesapeake", "rmat12_w7", "rmat14_w7"):
        g = golden[name]
        Ap, Aj, Ax = golden_graph(oracle, g)
        ap, aj, ax = dev(Ap), dev(Aj), dev(Ax)
        for run in g["runs"]:
            w = torch.empty(len(Ap) - 1, dtype=torch.float32, device="cuda")
            rc.sssp(ap, aj, ax, run["source"], w)
            assert sha(w.cpu().numpy().view(np.uint32)) == run["sssp_bits_sha256"], (name, run["source"])
            d = torch.empty(len(Ap) - 1, dtype=torch.int32, device="cuda")
            rc.bfs(ap, aj, ax, run["source"], d)
            assert sha(d.cpu().numpy()) == run["bfs_sha256"], (name, run["source"])


@pytest.mark.parametrize("algo", ["bfs", "sssp", "pr"])
def test_reference_example_harness(algo, oracle):
    """The reference's own example harness (examples/algorithms/<algo>/<algo>.cu, the program its CI
    runs on chesapeake: .github/workflows/ubuntu.yml:52-79), compiled in place and unmodified
    against include/gunrock (oracle/ref_build.sh), run on the golden chesapeake.mtx.  bfs / sssp
    compare the engine's result with the reference's CPU checker inside the harness and must print
    `Number of errors : 0`; pr has no checker there (pr.cu:64-70): its printed head is compared with
    the oracle's restatement."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(GOLDEN_DIR))
    exe = os.path.join(repo, "oracle", "_ref", "ref_" + algo)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/ref_{algo} not built (reference tree was not mounted)")
    mtx = os.path.join(GOLDEN_DIR, "chesapeake.mtx")
    r = subprocess.run([exe, mtx], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = r.stdout
    assert "GPU Elapsed Time" in out
    n, Ap, Aj, Ax = oracle.mtx_to_csr(mtx)
    if algo == "pr":
        line = [l for l in out.splitlines() if l.startswith("GPU rank[:")][0]
        got = np.array([float(x) for x in line.split("=")[1].split()], dtype=np.float32)
        want, _ = oracle.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
        assert len(got) == min(40, n) and np.abs(got - want[:len(got)]).max() < 1e-5
        return
    assert "Number of errors : 0" in out, out[-1500:]
    gpu = [l for l in out.splitlines() if l.startswith("GPU distances[:")][0]
    got = [float(x) for x in gpu.split("=")[1].split()]
    want = (oracle.bfs_heap(Ap, Aj, 0)[0] if algo == "bfs" else oracle.sssp_heap(Ap, Aj, Ax, 0)[0])
    assert got == [float(x) for x in want[:len(got)]]


def test_reference_headers_as_a_partitioned_job_of_one_on_rccl(refc, oracle, golden):
    """bfs.hxx / sssp.hxx UNCHANGED with the context attached to an RCCL job: enactor_t::enact()
    exchanges the frontier between supersteps (framework/partitioned.hxx; the combiner of
    `result.distances` is declared outside the client headers).  One GPU hosts one RCCL rank, so
    this is the production call sequence (ncclCommInitRank, ncclAllGather on the engine's stream)
    with a single participant; several ranks run it over host callbacks in test_gpu_distributed.
    pr.hxx declares no combiner: a partitioned run is refused ("replicas only")."""
    import torch
    import essentials_amd as ea
    if not hasattr(refc.L, "refc_bfs_job"):
        pytest.skip("libgrx_ref_clients.so predates the job entry points")
    for name in ("chesapeake", "sample4x4", "rmat10_w7", "rmat14_w0", "rmat10_directed"):
        g = golden[name]
        Ap, Aj, Ax = golden_graph(oracle, g)
        ap, aj, ax = dev(Ap), dev(Aj), dev(Ax)
        n = len(Ap) - 1
        for run in g["runs"]:
            d = torch.empty(n, dtype=torch.int32, device="cuda")
            refc.run_job("bfs", ap, aj, ax, run["source"], d, 0, 1, 0, n, unique_id=ea.Context.unique_id())
            assert sha(d.cpu().numpy()) == run["bfs_sha256"], (name, run["source"])
            w = torch.empty(n, dtype=torch.float32, device="cuda")
            refc.run_job("sssp", ap, aj, ax, run["source"], w, 0, 1, 0, n, unique_id=ea.Context.unique_id())
            assert sha(w.cpu().numpy().view(np.uint32)) == run["sssp_bits_sha256"], (name, run["source"])
    n, Ap, Aj, Ax = oracle.rmat_csr(9, 8, 3, 0, False)
    p = torch.empty(n, dtype=torch.float32, device="cuda")
    assert refc.pr_job_refused(dev(Ap), dev(Aj), dev(Ax), p, ea.Context.unique_id())


@pytest.mark.parametrize("algo", ["kcore", "ppr", "color", "spmv"])
def test_reference_neighbour_harnesses(algo):
    """Beyond the hot path's three clients: the reference's kcore, ppr and color example harnesses (and
    their algorithm headers), compiled in place and unmodified against include/gunrock, checked by
    the reference's OWN CPU implementations inside the harness (kcore_cpu.hxx, ppr_cpu.hxx).  They
    lean on the operators next to advance -- predicated filters whose predicates have side effects
    (kcore.hxx:150-175, ppr.hxx:120-145), parallel_for, batch over host threads, frontier
    sequence() -- SURVEY.md 8(f) rows f2 / f3; spmv runs its PULL form, i.e.
    operators::neighborreduce (spmv.hxx:107-128)."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(GOLDEN_DIR))
    exe = os.path.join(repo, "oracle", "_ref", "ref_" + algo)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/ref_{algo} not built (reference tree was not mounted)")
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "chesapeake.mtx")], capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Number of errors : 0" in r.stdout, r.stdout[-1500:]


def test_reference_bc_harness_against_brandes(oracle):
    """bc.hxx (merge_path advances over explicit per-depth frontiers, vertices -> none, one job per
    source through operators::batch) has no checker in the reference; its printed values are
    compared with Brandes' algorithm run here on the same graph (every ordered pair counted, i.e.
    twice the undirected textbook value)."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(GOLDEN_DIR))
    exe = os.path.join(repo, "oracle", "_ref", "ref_bc")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_bc not built (reference tree was not mounted)")
    mtx = os.path.join(GOLDEN_DIR, "chesapeake.mtx")
    r = subprocess.run([exe, mtx], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("GPU bc values[:")][0]
    got = np.array([float(x) for x in line.split("=")[1].split()])
    n, Ap, Aj, _ = oracle.mtx_to_csr(mtx)
    bc = np.zeros(n)
    for s in range(n):                      # Brandes, unweighted
        sigma = np.zeros(n); sigma[s] = 1
        dist = np.full(n, -1); dist[s] = 0
        order, preds, q = [], [[] for _ in range(n)], [s]
        while q:
            nq = []
            for u in q:
                order.append(u)
                for v in Aj[Ap[u]:Ap[u + 1]]:
                    if dist[v] < 0:
                        dist[v] = dist[u] + 1
                        nq.append(v)
                    if dist[v] == dist[u] + 1:
                        sigma[v] += sigma[u]
                        preds[v].append(u)
            q = list(dict.fromkeys(nq))
        delta = np.zeros(n)
        for w in reversed(order):
            for u in preds[w]:
                delta[u] += sigma[u] / sigma[w] * (1 + delta[w])
            if w != s:
                bc[w] += delta[w]
    k = len(got)
    scale = 1.0 if np.allclose(got, bc[:k], rtol=1e-3, atol=1e-3) else 0.5
    assert np.allclose(got, bc[:k] * scale, rtol=1e-3, atol=1e-3), (got[:8], bc[:8])


def test_reference_tc_known_answers(refc):
    """The reference's only pinned algorithm results (unittests/algorithms/tc.cuh:19-93): its
    unchanged tc.hxx -- a block_mapped graph -> none advance whose functor intersects neighbour
    lists -- on this engine must give the triangle counts the reference's unit tests expect."""
    import torch
    if not hasattr(refc.L, "refc_tc"):
        pytest.skip("libgrx_ref_clients.so predates the tc entry point")
    for Ap, Aj in (([0, 3, 5, 8, 10], [1, 2, 3, 0, 2, 0, 1, 3, 0, 2]),                 # tc.cuh:22-31
                   ([0, 4, 7, 10, 12], [0, 1, 2, 3, 0, 1, 2, 0, 1, 3, 0, 2])):        # :59-70 (self loop)
        ap = dev(np.array(Ap, np.int32))
        aj = dev(np.array(Aj, np.int32))
        ax = dev(np.zeros(len(Aj), np.float32))
        counts = torch.zeros(4, dtype=torch.int32, device="cuda")
        total = refc.tc(ap, aj, ax, counts)
        assert counts.cpu().tolist() == [2, 1, 2, 1] and total == 6           # tc.cuh:46-55, :85-93
