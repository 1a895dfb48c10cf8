"""ctypes binding of include/essentials_amd.h and a small object layer over it.

Names follow the reference's domain (graph, frontier, advance, filter, enactor stats).
Mirrors, argument for argument, ``gunrock::{bfs,sssp,pr}::run`` (reference
algorithms/bfs.hxx:151-176, sssp.hxx:155-185, pr.hxx:182-216) and the frontier-level
operator overloads (advance.hxx:91-129, filter.hxx:59-86, uniquify.hxx:15-42).
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# ESSENTIALS_AMD_LIB selects an experiment build of the SAME library (see build.build_variant)
_LIB_PATH = os.environ.get("ESSENTIALS_AMD_LIB") or os.path.join(_PKG, "libessentials_amd.so")

INT_UNREACHED = 2**31 - 1
FLT_UNREACHED = float(np.finfo(np.float32).max)


class EngineError(RuntimeError):
    """A C-ABI call failed (the C++ surface threw gunrock::error::exception_t).  `code` is the
    grx_status: -1 invalid argument, -2 runtime, -3 unsupported, -4 a peer rank's superstep failed,
    -5 a superstep's gather timed out (exit without synchronising the context)."""
    code = 0


ERR_PEER, ERR_TIMEOUT = -4, -5


class LoadBalance(enum.IntEnum):
    thread_mapped = 0
    warp_mapped = 1
    block_mapped = 2
    bucketing = 3
    merge_path = 4
    merge_path_v2 = 5
    work_stealing = 6


class FilterAlgorithm(enum.IntEnum):
    remove = 0
    predicated = 1
    compact = 2
    bypass = 3


class UniquifyAlgorithm(enum.IntEnum):
    unique = 0
    unique_copy = 1


class EdgeOp(enum.IntEnum):
    all = 0
    bfs = 1
    sssp = 2
    count_edge = 3
    sum_weight = 4


class VertexOp(enum.IntEnum):
    all = 0
    odd = 1
    once = 2
    count = 3


class _Options(C.Structure):
    _fields_ = [("load_balance", C.c_int32), ("holes_layout", C.c_int32),
                ("hub_threshold", C.c_int32), ("max_iterations", C.c_int32),
                ("frontier_sizing_factor", C.c_float), ("collect_kernel_time", C.c_int32),
                ("chunk_edges", C.c_int32), ("direction_optimized", C.c_int32),
                ("do_alpha", C.c_float), ("do_beta", C.c_float),
                ("chunk_queue_limit", C.c_int32), ("sssp_two_pass", C.c_int32),
                ("call_every_edge", C.c_int32)]


class _Stats(C.Structure):
    _fields_ = [("elapsed_ms", C.c_float), ("advance_kernel_ms", C.c_float),
                ("iterations", C.c_int32), ("advance_launches", C.c_int32),
                ("vertices_reached", C.c_int64), ("edges_traversed", C.c_int64),
                ("levels_recorded", C.c_int32), ("pull_iterations", C.c_int32),
                ("frontier_slots", C.c_int64 * 64), ("edges_expanded", C.c_int64)]


class _PartitionedStats(C.Structure):
    _fields_ = [("elapsed_ms", C.c_float), ("supersteps", C.c_int32), ("collectives", C.c_int32),
                ("bitmap_supersteps", C.c_int32), ("allreduce_supersteps", C.c_int32),
                ("iterations", C.c_int32), ("last_error", C.c_float),
                ("pairs_exchanged", C.c_int64), ("bytes_sent", C.c_int64),
                ("large_gather_supersteps", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


# collective callbacks of grx_context_attach_collectives
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32,
                            C.c_void_p)
UNIQUE_ID_BYTES = 128


@dataclass
class Options:
    load_balance: LoadBalance = LoadBalance.block_mapped
    holes_layout: bool = False
    hub_threshold: int = 0
    max_iterations: int = 0
    frontier_sizing_factor: float = 1.5
    collect_kernel_time: bool = False
    chunk_edges: int = 0
    direction_optimized: bool = False
    do_alpha: float = 0.0
    do_beta: float = 0.0
    chunk_queue_limit: int = 0
    sssp_two_pass: bool = False
    call_every_edge: bool = False

    def _c(self) -> _Options:
        o = _Options()
        o.load_balance = int(self.load_balance)
        o.holes_layout = int(self.holes_layout)
        o.hub_threshold = int(self.hub_threshold)
        o.max_iterations = int(self.max_iterations)
        o.frontier_sizing_factor = float(self.frontier_sizing_factor)
        o.collect_kernel_time = int(self.collect_kernel_time)
        o.chunk_edges = int(self.chunk_edges)
        o.direction_optimized = int(self.direction_optimized)
        o.do_alpha = float(self.do_alpha)
        o.do_beta = float(self.do_beta)
        o.chunk_queue_limit = int(self.chunk_queue_limit)
        o.sssp_two_pass = int(self.sssp_two_pass)
        o.call_every_edge = int(self.call_every_edge)
        return o


@dataclass
class Stats:
    elapsed_ms: float = 0.0
    advance_kernel_ms: float = 0.0
    iterations: int = 0
    advance_launches: int = 0
    vertices_reached: int = 0
    edges_traversed: int = 0
    frontier_slots: list = field(default_factory=list)
    pull_iterations: int = 0
    edges_expanded: int = 0

    @staticmethod
    def _from(s: _Stats) -> "Stats":
        return Stats(s.elapsed_ms, s.advance_kernel_ms, s.iterations, s.advance_launches,
                     s.vertices_reached, s.edges_traversed,
                     list(s.frontier_slots[: s.levels_recorded]), s.pull_iterations,
                     s.edges_expanded)


# symbol -> (restype, argtypes); the list is also what tests check against the header
_VP = C.c_void_p
_SIGNATURES = {
    "grx_abi_version": (C.c_int, []),
    "grx_last_error": (C.c_char_p, []),
    "grx_default_options": (None, [C.POINTER(_Options)]),
    "grx_context_create": (C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    "grx_context_destroy": (C.c_int, [_VP]),
    "grx_context_synchronize": (C.c_int, [_VP]),
    "grx_context_wait_stream": (C.c_int, [_VP, _VP]),
    "grx_trim_cache": (C.c_int, []),
    "grx_copy_to_host": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "grx_copy_to_device": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "grx_job_unique_id": (C.c_int, [_VP]),
    "grx_context_attach_rccl": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "grx_context_attach_collectives": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    "grx_context_detach": (C.c_int, [_VP]),
    "grx_context_job_info": (C.c_int, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p,
                                       C.c_size_t]),
    "grx_partitioned_create": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.POINTER(_Options),
                                         C.c_int64, C.c_int64, C.c_int64, C.POINTER(_VP)]),
    "grx_partitioned_destroy": (C.c_int, [_VP]),
    "grx_partitioned_run": (C.c_int, [_VP, C.c_int32, C.c_int32, _VP, C.POINTER(_PartitionedStats)]),
    "grx_partitioned_pagerank": (C.c_int, [_VP, C.c_float, C.c_float, C.c_int32, _VP,
                                           C.POINTER(_PartitionedStats)]),
    "grx_context_device_info": (C.c_int, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]),
    "grx_graph_from_device_csr": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP,
                                            C.POINTER(_VP)]),
    "grx_graph_from_host_csr": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP,
                                          C.POINTER(_VP)]),
    "grx_graph_from_mtx": (C.c_int, [C.c_char_p, C.POINTER(_VP)]),
    "grx_graph_from_csr_file": (C.c_int, [C.c_char_p, C.POINTER(_VP)]),
    "grx_graph_write_csr_file": (C.c_int, [_VP, C.c_char_p]),
    "grx_graph_rmat": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                 C.POINTER(_VP)]),
    "grx_graph_sorted_rows": (C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    "grx_graph_build_in_edges": (C.c_int, [_VP, _VP]),
    "grx_graph_hot_first": (C.c_int, [_VP, _VP, C.c_int]),
    "grx_graph_destroy": (C.c_int, [_VP]),
    "grx_graph_info": (C.c_int, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int64), C.POINTER(_VP), C.POINTER(_VP),
                                 C.POINTER(_VP)]),
    "grx_graph_copy_to_host": (C.c_int, [_VP, _VP, _VP, _VP]),
    "grx_bfs": (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP, C.POINTER(_Options), C.POINTER(_Stats)]),
    "grx_sssp": (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP, C.POINTER(_Options), C.POINTER(_Stats)]),
    "grx_pagerank": (C.c_int, [_VP, _VP, C.c_float, C.c_float, _VP, C.POINTER(_Options),
                               C.POINTER(_Stats)]),
    "grx_advance": (C.c_int, [_VP, _VP, C.POINTER(_Options), C.c_int32, _VP, C.c_int32, _VP,
                              C.c_int64, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "grx_filter": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, _VP, C.c_int32, _VP, C.c_int64, _VP,
                             C.c_int64, C.POINTER(C.c_int64)]),
    "grx_uniquify": (C.c_int, [_VP, C.c_int32, C.c_int32, _VP, C.c_int64, _VP, C.c_int64,
                               C.POINTER(C.c_int64)]),
    "grx_graph_partition": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(_VP), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32)]),
    "grx_graph_partition_hot_first": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(_VP),
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "grx_partitioned_expand": (C.c_int, [_VP, _VP, C.POINTER(_Options), C.c_int32, _VP, C.c_int32,
                                         _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, C.c_int64,
                                         C.POINTER(C.c_int64)]),
    "grx_partitioned_level_bitmap": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, _VP, C.c_int64]),
    "grx_partitioned_admit": (C.c_int, [_VP, C.c_int32, _VP, C.c_int64, _VP, C.c_int32, _VP,
                                        C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_int32, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "grx_partitioned_step": (C.c_int, [_VP, _VP, C.POINTER(_Options), C.c_int32, _VP, _VP, _VP,
                                       C.c_int32, _VP, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                       C.c_int32, C.c_int32, _VP, C.c_int64, _VP, _VP, C.c_int64,
                                       _VP, C.c_int64, _VP]),
    "grx_pagerank_partitioned_scatter": (C.c_int, [_VP, _VP, C.c_float, _VP, _VP, C.c_int32, _VP,
                                                   C.c_int32, C.c_int32, C.POINTER(_Options)]),
    "grx_measure_copy_bandwidth": (C.c_int, [_VP, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "grx_measure_gather_rate": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(C.c_double)]),
}

_lib = None


def library_path() -> str:
    return _LIB_PATH


def load_library():
    """Load libessentials_amd.so; there is no fallback of any kind."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise EngineError(
            f"{_LIB_PATH} is missing: build it with `python -m essentials_amd.build` "
            "(hipcc --offload-arch=gfx950). essentials_amd has no CPU path.")
    # torch ships its own copy of the HIP runtime (same SONAME as /opt/rocm's): load torch's
    # first so that this process has ONE runtime -- with the order reversed torch later fails
    # with "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.grx_abi_version() != 1:
        raise EngineError("ABI version mismatch")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().grx_last_error()
        err = EngineError(f"{what} failed ({rc}): {msg.decode(errors='replace') if msg else ''}")
        err.code = rc
        raise err


def _ptr(t) -> Optional[int]:
    """Device (torch) or host (numpy) pointer of an array-like, None -> NULL."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


class Context:
    """gcuda::multi_context_t(device[, stream])  (reference cuda/context.hxx:136-206)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._h = _VP()
        _check(load_library().grx_context_create(device, stream, C.byref(self._h)),
               "grx_context_create")
        self.device = device
        self._stream = stream   # None: the engine's private non-blocking stream

    def synchronize(self) -> None:
        _check(load_library().grx_context_synchronize(self._h), "grx_context_synchronize")

    def after_torch(self) -> None:
        """Order the engine's next work after everything torch has enqueued on ITS current stream
        of this device (event record + stream wait on the device; the host does not wait).  A
        default Context runs on a private non-blocking stream that is ordered after nothing:
        tensors a caller has just filled (`torch.full`, `clone`, `copy_`) must not be read by an
        engine kernel before the fill has run.  Every wrapper below that takes tensors calls this;
        it is a no-op when the context was created on torch's current stream."""
        import torch
        if not torch.cuda.is_available():
            return
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if self._stream is not None and cur == self._stream:
            return
        _check(load_library().grx_context_wait_stream(self._h, cur), "grx_context_wait_stream")

    def device_info(self) -> dict:
        cus, wave, mem = C.c_int32(), C.c_int32(), C.c_int64()
        name = C.create_string_buffer(256)
        _check(load_library().grx_context_device_info(self._h, cus, wave, mem, name, 256),
               "grx_context_device_info")
        return {"name": name.value.decode(), "compute_units": cus.value,
                "wavefront_size": wave.value, "total_memory_bytes": mem.value}

    def copy_bandwidth_gbps(self, nbytes: int = 1 << 30, repeats: int = 10) -> float:
        g = C.c_double()
        _check(load_library().grx_measure_copy_bandwidth(self._h, nbytes, repeats, g),
               "grx_measure_copy_bandwidth")
        return g.value

    # -- multi-GPU job (gcuda::multi_context_t::attach_job) ---------------------------------
    def attach_rccl(self, rank: int, world: int, unique_id: bytes) -> None:
        """Join an RCCL job: ncclCommInitRank with the id rank 0 got from Context.unique_id().
        Collective over all ranks."""
        assert len(unique_id) == UNIQUE_ID_BYTES
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        _check(load_library().grx_context_attach_rccl(self._h, rank, world, buf),
               "grx_context_attach_rccl")

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        _check(load_library().grx_job_unique_id(buf), "grx_job_unique_id")
        return buf.raw

    def attach_collectives(self, rank: int, world: int, all_gather, all_reduce) -> None:
        """Join a job whose collectives are host callbacks:
        all_gather(d_send: int, d_recv: int, bytes_per_rank: int, stream: int) -> int (0 = ok) and
        all_reduce(d_buffer: int, count: int, dtype: int, op: int, stream: int) -> int."""
        def _ag(_user, send, recv, nbytes, stream):
            try:
                return int(all_gather(send, recv, nbytes, stream) or 0)
            except Exception as e:   # an exception must not unwind through the C frames
                print(f"[essentials_amd] all_gather callback failed: {e!r}", flush=True)
                return -1

        def _ar(_user, buf, count, dtype, op, stream):
            try:
                return int(all_reduce(buf, count, dtype, op, stream) or 0)
            except Exception as e:
                print(f"[essentials_amd] all_reduce callback failed: {e!r}", flush=True)
                return -1
        self._callbacks = (ALL_GATHER_FN(_ag), ALL_REDUCE_FN(_ar))   # keep them alive
        _check(load_library().grx_context_attach_collectives(
            self._h, rank, world, C.cast(self._callbacks[0], _VP), C.cast(self._callbacks[1], _VP),
            None), "grx_context_attach_collectives")

    def detach(self) -> None:
        _check(load_library().grx_context_detach(self._h), "grx_context_detach")
        self._callbacks = None

    def job_info(self) -> dict:
        r, w = C.c_int32(), C.c_int32()
        name = C.create_string_buffer(32)
        _check(load_library().grx_context_job_info(self._h, r, w, name, 32), "grx_context_job_info")
        return {"rank": r.value, "world_size": w.value, "backend": name.value.decode()}

    def copy_to_host(self, h_dst: np.ndarray, d_src: int) -> None:
        _check(load_library().grx_copy_to_host(self._h, h_dst.ctypes.data, d_src, h_dst.nbytes),
               "grx_copy_to_host")

    def copy_to_device(self, d_dst: int, h_src: np.ndarray) -> None:
        _check(load_library().grx_copy_to_device(self._h, d_dst, h_src.ctypes.data, h_src.nbytes),
               "grx_copy_to_device")

    @staticmethod
    def trim_cache() -> None:
        """Return the device blocks the engine parked for reuse to the device (see
        grx_trim_cache): for hosts whose OTHER allocator (torch) runs short."""
        _check(load_library().grx_trim_cache(), "grx_trim_cache")

    def gather_rate(self, graph, mode: int = 1, repeats: int = 5) -> float:
        """Random-gather ceiling of `graph` in lookups (= edges) per second: table[column[i]] over
        all edges; mode 1 = the agent-scope loads atomic::min pre-tests with, 0 = plain loads."""
        r = C.c_double()
        _check(load_library().grx_measure_gather_rate(self._h, graph._h, mode, repeats, r),
               "grx_measure_gather_rate")
        return r.value

    def close(self) -> None:
        if self._h:
            load_library().grx_context_destroy(self._h)
            self._h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PartitionedPlan:
    """grx_partitioned_t: the persistent state of vertex-partitioned traversals on one rank and the
    C++ superstep loop over it (grx_partitioned_run / grx_partitioned_pagerank).  The context must
    already be attached to its job (Context.attach_rccl / attach_collectives)."""

    def __init__(self, ctx: Context, local: "Graph", row_begin: int, row_end: int,
                 options: Optional["Options"] = None, small_slot: int = 0, dense_threshold: int = 0,
                 replica_threshold: int = 0):
        self._h = _VP()
        self.ctx, self.local = ctx, local     # keep both alive
        o = (options or Options())._c()
        _check(load_library().grx_partitioned_create(ctx._h, local._h, row_begin, row_end, C.byref(o),
                                                     small_slot, dense_threshold, replica_threshold,
                                                     C.byref(self._h)), "grx_partitioned_create")

    def run(self, op: "EdgeOp", source: int, labels) -> dict:
        """labels: replica [V] (int32 for BFS, float32 for SSSP), overwritten.  Collective."""
        s = _PartitionedStats()
        self.ctx.after_torch()
        _check(load_library().grx_partitioned_run(self._h, int(op), source, _ptr(labels), C.byref(s)),
               "grx_partitioned_run")
        return s.as_dict()

    def pagerank(self, p, alpha: float = 0.85, tol: float = 1e-6, max_iterations: int = 0) -> dict:
        s = _PartitionedStats()
        self.ctx.after_torch()
        _check(load_library().grx_partitioned_pagerank(self._h, alpha, tol, max_iterations, _ptr(p),
                                                       C.byref(s)), "grx_partitioned_pagerank")
        return s.as_dict()

    def close(self) -> None:
        if self._h:
            load_library().grx_partitioned_destroy(self._h)
            self._h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """graph::graph_t over CSR (reference graph/build.hxx:26-36); int32 ids, float weights."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keep = keepalive
        n, m, nnz = C.c_int32(), C.c_int32(), C.c_int64()
        _check(load_library().grx_graph_info(self._h, n, m, nnz, None, None, None), "grx_graph_info")
        self.n_rows, self.n_cols, self.nnz = n.value, m.value, nnz.value

    @staticmethod
    def from_host_csr(row_offsets, col, val, n_cols=None) -> "Graph":
        ap = np.ascontiguousarray(row_offsets, np.int32)
        aj = np.ascontiguousarray(col, np.int32)
        ax = np.ascontiguousarray(val, np.float32)
        h = _VP()
        n = len(ap) - 1
        _check(load_library().grx_graph_from_host_csr(
            n, n if n_cols is None else n_cols, len(aj), ap.ctypes.data,
            aj.ctypes.data if len(aj) else None, ax.ctypes.data if len(ax) else None, C.byref(h)),
            "grx_graph_from_host_csr")
        return Graph(h)

    @staticmethod
    def from_device_csr(row_offsets, col, val) -> "Graph":
        """Non-owning view over torch int32/int32/float32 device tensors (kept alive here)."""
        h = _VP()
        n = row_offsets.numel() - 1
        # the handle reduces the largest degree from the arrays on first use: they must be complete
        _torch().cuda.current_stream(row_offsets.device).synchronize()
        _check(load_library().grx_graph_from_device_csr(n, n, col.numel(), _ptr(row_offsets),
                                                        _ptr(col), _ptr(val), C.byref(h)),
               "grx_graph_from_device_csr")
        return Graph(h, keepalive=(row_offsets, col, val))

    @staticmethod
    def from_mtx(path: str) -> "Graph":
        h = _VP()
        _check(load_library().grx_graph_from_mtx(path.encode(), C.byref(h)), "grx_graph_from_mtx")
        return Graph(h)

    @staticmethod
    def from_csr_file(path: str) -> "Graph":
        h = _VP()
        _check(load_library().grx_graph_from_csr_file(path.encode(), C.byref(h)),
               "grx_graph_from_csr_file")
        return Graph(h)

    @staticmethod
    def rmat(ctx: Context, scale: int, edge_factor: int = 16, seed: int = 1, weight_seed: int = 0,
             symmetrize: bool = True) -> "Graph":
        h = _VP()
        _check(load_library().grx_graph_rmat(ctx._h, scale, edge_factor, seed, weight_seed,
                                             int(symmetrize), C.byref(h)), "grx_graph_rmat")
        return Graph(h)

    def sorted_rows(self, ctx: "Context") -> "Graph":
        """Same graph, neighbour lists sorted by column id (SuiteSparse-style layout)."""
        h = _VP()
        _check(load_library().grx_graph_sorted_rows(ctx._h, self._h, C.byref(h)),
               "grx_graph_sorted_rows")
        return Graph(h)

    def build_in_edges(self, ctx: "Context") -> "Graph":
        """Attach the transpose (in-edges) of a DIRECTED graph: enables pull / direction-optimised BFS."""
        _check(load_library().grx_graph_build_in_edges(ctx._h, self._h), "grx_graph_build_in_edges")
        return self

    def hot_first(self, ctx: "Context", enable: bool = True) -> "Graph":
        """Build now (or drop and disable) the hot-first renumbered copy bfs / sssp run on; labels
        always come back in this graph's own numbering (include/essentials_amd.h)."""
        _check(load_library().grx_graph_hot_first(ctx._h, self._h, int(enable)), "grx_graph_hot_first")
        return self

    def write_csr_file(self, path: str) -> None:
        _check(load_library().grx_graph_write_csr_file(self._h, path.encode()),
               "grx_graph_write_csr_file")

    def to_host(self):
        ap = np.empty(self.n_rows + 1, np.int32)
        aj = np.empty(max(self.nnz, 1), np.int32)
        ax = np.empty(max(self.nnz, 1), np.float32)
        _check(load_library().grx_graph_copy_to_host(self._h, ap.ctypes.data, aj.ctypes.data,
                                                     ax.ctypes.data), "grx_graph_copy_to_host")
        return ap, aj[: self.nnz], ax[: self.nnz]

    def offsets_to_host(self):
        """Row offsets only (4 (V + 1) bytes; to_host() moves 8 bytes per edge as well)."""
        ap = np.empty(self.n_rows + 1, np.int32)
        _check(load_library().grx_graph_copy_to_host(self._h, ap.ctypes.data, None, None),
               "grx_graph_copy_to_host")
        return ap

    def close(self) -> None:
        if self._h:
            load_library().grx_graph_destroy(self._h)
            self._h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch():
    import torch
    return torch


def bfs(ctx: Context, g: Graph, source: int, distances=None, options: Optional[Options] = None):
    """gunrock::bfs::run -> (int32 depths on the device, Stats)."""
    torch = _torch()
    if distances is None:
        distances = torch.empty(g.n_rows, dtype=torch.int32, device=f"cuda:{ctx.device}")
    o = (options or Options())._c()
    s = _Stats()
    ctx.after_torch()
    _check(load_library().grx_bfs(ctx._h, g._h, source, _ptr(distances), None, C.byref(o),
                                  C.byref(s)), "grx_bfs")
    return distances, Stats._from(s)


def sssp(ctx: Context, g: Graph, source: int, distances=None, options: Optional[Options] = None):
    """gunrock::sssp::run -> (float32 distances on the device, Stats)."""
    torch = _torch()
    if distances is None:
        distances = torch.empty(g.n_rows, dtype=torch.float32, device=f"cuda:{ctx.device}")
    o = (options or Options())._c()
    s = _Stats()
    ctx.after_torch()
    _check(load_library().grx_sssp(ctx._h, g._h, source, _ptr(distances), None, C.byref(o),
                                   C.byref(s)), "grx_sssp")
    return distances, Stats._from(s)


def pagerank(ctx: Context, g: Graph, alpha: float = 0.85, tol: float = 1e-6, p=None,
             options: Optional[Options] = None):
    """gunrock::pr::run -> (float32 ranks on the device, Stats)."""
    torch = _torch()
    if p is None:
        p = torch.empty(g.n_rows, dtype=torch.float32, device=f"cuda:{ctx.device}")
    o = (options or Options())._c()
    s = _Stats()
    ctx.after_torch()
    _check(load_library().grx_pagerank(ctx._h, g._h, alpha, tol, _ptr(p), C.byref(o), C.byref(s)),
           "grx_pagerank")
    return p, Stats._from(s)


def advance(ctx: Context, g: Graph, frontier, op: EdgeOp = EdgeOp.all, state=None, iparam: int = 0,
            options: Optional[Options] = None, want_output: bool = True,
            capacity: Optional[int] = None):
    """operators::advance::execute (frontier overload).  frontier=None -> whole graph.

    Returns the output frontier (int32 device tensor, order unspecified) or None.
    """
    torch = _torch()
    o = (options or Options())._c()
    n_in = 0 if frontier is None else frontier.numel()
    if frontier is not None and n_in == 0:
        # an EMPTY frontier is not "the whole graph": hand the ABI a non-NULL pointer
        frontier = torch.empty(1, dtype=torch.int32, device=f"cuda:{ctx.device}")
    out = None
    cap = 0
    if want_output:
        cap = capacity if capacity is not None else max(int(g.nnz) * 2 + 16, 16)
        out = torch.empty(cap, dtype=torch.int32, device=f"cuda:{ctx.device}")
    n_out = C.c_int64()
    ctx.after_torch()
    _check(load_library().grx_advance(ctx._h, g._h, C.byref(o), int(op), _ptr(state), iparam,
                                      _ptr(frontier), n_in, _ptr(out), cap, C.byref(n_out)),
           "grx_advance")
    return out[: n_out.value] if want_output else None


def filter(ctx: Context, g: Graph, frontier, algorithm: FilterAlgorithm,
           pred: VertexOp = VertexOp.all, state=None, iparam: int = 0):
    """operators::filter::execute (frontier overload) -> new frontier tensor."""
    torch = _torch()
    n_in = frontier.numel()
    out = torch.empty(max(n_in, 1), dtype=torch.int32, device=f"cuda:{ctx.device}")
    n_out = C.c_int64()
    ctx.after_torch()
    _check(load_library().grx_filter(ctx._h, g._h, int(algorithm), int(pred), _ptr(state), iparam,
                                     _ptr(frontier) if n_in else None, n_in, _ptr(out),
                                     out.numel(), C.byref(n_out)), "grx_filter")
    return out[: n_out.value]


def uniquify(ctx: Context, frontier, algorithm: UniquifyAlgorithm = UniquifyAlgorithm.unique,
             best_effort: bool = False):
    """operators::uniquify::execute (frontier overload) -> deduplicated frontier tensor."""
    torch = _torch()
    n_in = frontier.numel()
    work = frontier.clone()
    out = torch.empty(max(n_in, 1), dtype=torch.int32, device=f"cuda:{ctx.device}")
    n_out = C.c_int64()
    ctx.after_torch()
    _check(load_library().grx_uniquify(ctx._h, int(algorithm), int(best_effort),
                                       _ptr(work) if n_in else None, n_in, _ptr(out), out.numel(),
                                       C.byref(n_out)), "grx_uniquify")
    res = work if algorithm == UniquifyAlgorithm.unique else out
    return res[: n_out.value]
