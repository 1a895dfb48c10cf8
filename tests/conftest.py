"""pytest configuration: markers, shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU).
`-m gpu`      : parity of the HIP path (through the C-ABI) against the oracle.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return {g["name"]: g for g in json.load(f)["graphs"]}


def golden_graph(oracle, g):
    """Rebuild the CSR a golden entry describes (full arrays, the mtx file, or the RMAT spec)."""
    if "row_offsets" in g:
        return (np.array(g["row_offsets"], np.int32), np.array(g["col"], np.int32),
                np.array(g["val"], np.float32))
    if g["name"] == "chesapeake":
        _, Ap, Aj, Ax = oracle.mtx_to_csr(os.path.join(GOLDEN_DIR, "chesapeake.mtx"))
        return Ap, Aj, Ax
    _, Ap, Aj, Ax = oracle.rmat_csr(g["scale"], g["edge_factor"], g["seed"], g["weight_seed"],
                                    g.get("symmetrize", True))
    return Ap, np.ascontiguousarray(Aj), np.ascontiguousarray(Ax)
