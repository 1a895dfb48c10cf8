"""Runners used by bench.py and the multi-process tests: one graph resident in HBM, repeated
BFS / SSSP traversals through the C ABI.

SingleRunner      one GPU, the whole graph.
PartitionedRunner one process per GPU (torch.distributed / RCCL): 1-D vertex partition, local
                  advance, all-gather of the per-rank output frontiers between supersteps
                  (SURVEY.md 8e).  Defined in this module once the RCCL path is built.
"""
from __future__ import annotations

import numpy as np

from . import api as ea

HBM_PEAK_GBPS = 8000.0


def bfs_algorithmic_bytes(edges_traversed: int, vertices_reached: int) -> int:
    """per traversed edge 4 B column index + 4 B label gather; per reached vertex
    4 B frontier read + 8 B row offsets + 4 B label write + 4 B next-frontier write."""
    return 8 * edges_traversed + 20 * vertices_reached


class SingleRunner:
    def __init__(self, ctx: ea.Context, scale: int, edge_factor: int, seed: int, weight_seed: int):
        import torch
        self.ctx = ctx
        self.g = ea.Graph.rmat(ctx, scale, edge_factor, seed, weight_seed, True)
        self.n, self.nnz = self.g.n_rows, self.g.nnz
        dev = f"cuda:{ctx.device}"
        self.depth = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.dist = torch.empty(self.n, dtype=torch.float32, device=dev)
        self._host = None
        self.last = {}

    def host_csr(self):
        if self._host is None:
            self._host = self.g.to_host()
        return self._host

    def global_degrees(self):
        return np.diff(self.host_csr()[0]).astype(np.int64)

    def bfs(self, source: int, opts: ea.Options) -> int:
        _, st = ea.bfs(self.ctx, self.g, source, self.depth, opts)
        self.last["bfs"] = st
        return st.edges_traversed

    def sssp(self, source: int, opts: ea.Options) -> int:
        _, st = ea.sssp(self.ctx, self.g, source, self.dist, opts)
        self.last["sssp"] = st
        return st.edges_traversed

    def bfs_roofline(self, source: int, lb) -> dict:
        """Kernel-level roofline of one BFS: HIP events around every advance launch."""
        best = None
        for _ in range(3):
            _, st = ea.bfs(self.ctx, self.g, source, self.depth,
                           ea.Options(load_balance=lb, collect_kernel_time=True))
            if best is None or st.advance_kernel_ms < best.advance_kernel_ms:
                best = st
        nbytes = bfs_algorithmic_bytes(best.edges_traversed, best.vertices_reached)
        achieved = nbytes / (best.advance_kernel_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "block_mapped_kernel+chunk_kernel (BFS advance, all levels of one traversal)",
                "algorithmic_bytes": nbytes, "kernel_ms": best.advance_kernel_ms,
                "launches": best.advance_launches, "enact_ms": best.elapsed_ms,
                "edges_traversed": best.edges_traversed, "vertices_reached": best.vertices_reached,
                "kernel_gteps": best.edges_traversed / (best.advance_kernel_ms * 1e-3) / 1e9}

    def detail(self) -> dict:
        d = {}
        for k, st in self.last.items():
            d[k] = {"enact_ms": st.elapsed_ms, "iterations": st.iterations,
                    "edges_traversed": st.edges_traversed, "vertices_reached": st.vertices_reached,
                    "mteps_enact": st.edges_traversed / max(st.elapsed_ms, 1e-9) / 1e3,
                    "frontier_slots": st.frontier_slots[:16]}
        return d
