#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REFERENCE's own CPU checkers.

Run in the build container only (needs /root/reference mounted so that
oracle/ref_build.sh can compile examples/algorithms/{bfs/bfs_cpu,sssp/sssp_cpu}.hxx
in place).  Only the vectors travel; the reference does not.

    python tests/golden/make_golden.py

Inputs:
  * chesapeake.mtx      -- the reference's in-tree dataset (datasets/chesapeake/),
                           a data file, committed next to this script
  * io::sample::csr()   -- the 4x4 fixture of include/gunrock/io/sample.hxx:58-93
                           (values typed in below as data)
  * tc.cuh graphs       -- unittests/algorithms/tc.cuh:19-93 CSR arrays (data)
  * RMAT scale 8..14    -- the build's own generator (oracle/grx_oracle.c), seeds fixed
Outputs per graph: CSR digests, BFS depths / SSSP distances computed by the
reference checkers (full arrays when small, sha256 + histogram otherwise).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.oracle import Oracle, RefOracle, build  # noqa: E402

INF_I = 2**31 - 1


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def describe(ref, name, Ap, Aj, Ax, sources, full):
    g = {"name": name, "n": int(len(Ap) - 1), "nnz": int(len(Aj)),
         "row_offsets_sha256": digest(Ap), "col_sha256": digest(Aj), "val_sha256": digest(Ax),
         "runs": []}
    if full:
        g["row_offsets"] = Ap.tolist(); g["col"] = Aj.tolist(); g["val"] = Ax.tolist()
    for s in sources:
        d, _ = ref.bfs(Ap, Aj, s)
        w, _ = ref.sssp(Ap, Aj, Ax, s)
        reached = d[d != INF_I]
        run = {"source": int(s),
               "bfs_sha256": digest(d), "sssp_bits_sha256": digest(w.view(np.uint32)),
               "reached": int(len(reached)), "max_depth": int(reached.max()),
               "depth_sum": int(reached.astype(np.int64).sum()),
               "depth_hist": np.bincount(reached).tolist(),
               "edges_traversed": int((Ap[1:] - Ap[:-1])[d != INF_I].sum()),
               "sssp_sum": float(w[w < 3e38].astype(np.float64).sum())}
        if full:
            run["bfs"] = d.tolist()
            run["sssp_bits"] = w.view(np.uint32).tolist()
        g["runs"].append(run)
    return g


def main():
    build()
    assert RefOracle.available(), "reference tree not mounted: cannot regenerate goldens"
    o, ref = Oracle(), RefOracle()
    graphs = []
    n, Ap, Aj, Ax = o.mtx_to_csr(os.path.join(HERE, "chesapeake.mtx"))
    graphs.append(describe(ref, "chesapeake", Ap, Aj, Ax, [0, 5, 38], True))
    # io::sample::csr()  (include/gunrock/io/sample.hxx:58-93)
    Ap = np.array([0, 0, 2, 3, 4], np.int32); Aj = np.array([0, 1, 2, 1], np.int32)
    Ax = np.array([5, 8, 3, 6], np.float32)
    graphs.append(describe(ref, "sample4x4", Ap, Aj, Ax, [1, 3], True))
    # unittests/algorithms/tc.cuh:19-36 and :57-74
    Ap = np.array([0, 3, 5, 8, 10], np.int32); Aj = np.array([1, 2, 3, 0, 2, 0, 1, 3, 0, 2], np.int32)
    graphs.append(describe(ref, "tc4", Ap, Aj, np.ones(10, np.float32), [0, 3], True))
    Ap = np.array([0, 4, 7, 10, 12], np.int32)
    Aj = np.array([0, 1, 2, 3, 0, 1, 2, 0, 1, 3, 0, 2], np.int32)
    graphs.append(describe(ref, "tc4_selfloop", Ap, Aj, np.ones(12, np.float32), [0], True))
    for scale, full in ((8, True), (10, False), (12, False), (14, False)):
        for wseed in (0, 7):
            n, Ap, Aj, Ax = o.rmat_csr(scale, 16, seed=1, weight_seed=wseed)
            deg = np.diff(Ap)
            srcs = [0, int(np.flatnonzero(deg > 0)[len(np.flatnonzero(deg > 0)) // 2])]
            g = describe(ref, f"rmat{scale}_w{wseed}", Ap, Aj, Ax, srcs, full and wseed == 0)
            g.update({"scale": scale, "edge_factor": 16, "seed": 1, "weight_seed": wseed,
                      "max_degree": int(deg.max()), "isolated": int((deg == 0).sum())})
            graphs.append(g)
    # a directed (non-symmetrised) one: sinks appear in frontiers
    n, Ap, Aj, Ax = o.rmat_csr(10, 8, seed=3, weight_seed=5, symmetrize=False)
    g = describe(ref, "rmat10_directed", Ap, Aj, Ax, [0, 1], False)
    g.update({"scale": 10, "edge_factor": 8, "seed": 3, "weight_seed": 5, "symmetrize": False})
    graphs.append(g)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "source_of_truth": "reference bfs_cpu.hxx:20-68 / sssp_cpu.hxx:22-72 compiled in place",
                   "graphs": graphs}, f, indent=1)
    print("wrote", len(graphs), "graphs")


if __name__ == "__main__":
    main()
