"""essentials_amd -- MI355X-native frontier advance/filter engine (host-side mirror).

The product is the HIP library ``libessentials_amd.so`` (C ABI: ``include/essentials_amd.h``)
plus the C++ header surface ``include/gunrock/``.  This package is the thin Python host
layer above the C ABI: device memory comes from torch tensors, streams from torch, process
groups from ``torch.distributed`` (RCCL).  There is NO CPU fallback: every call goes to the
HIP library and fails loudly when it is missing.
"""
from .api import (  # noqa: F401
    Context, Graph, Options, Stats, PartitionedPlan, LoadBalance, FilterAlgorithm, UniquifyAlgorithm, EdgeOp,
    VertexOp, EngineError, bfs, sssp, pagerank, advance, filter, uniquify, library_path,
    INT_UNREACHED, FLT_UNREACHED,
)

__all__ = [
    "Context", "Graph", "Options", "Stats", "PartitionedPlan", "LoadBalance", "FilterAlgorithm", "UniquifyAlgorithm",
    "EdgeOp", "VertexOp", "EngineError", "bfs", "sssp", "pagerank", "advance", "filter",
    "uniquify", "library_path", "INT_UNREACHED", "FLT_UNREACHED",
]
