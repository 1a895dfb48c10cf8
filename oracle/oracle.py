"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module (see oracle/grx_oracle.h).  The product package
``essentials_amd`` never does.

``Oracle``    : the C restatement (oracle/grx_oracle.c -> libgrx_oracle.so)
``RefOracle`` : the reference's own bfs_cpu.hxx / sssp_cpu.hxx compiled in place
                (oracle/ref_build.sh -> oracle/_ref/libgrx_ref_oracle.so); may be
                absent, ``RefOracle.available()`` says so.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgrx_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libgrx_ref_oracle.so")
_REF_CLIENTS = os.path.join(_HERE, "_ref", "libgrx_ref_clients.so")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


class Trace(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("edges_traversed", C.c_int64),
        ("frontier_slots", C.c_int64 * 64),
        ("frontier_valid", C.c_int64 * 64),
    ]


EDGE_OP = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p)
VERTEX_OP = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_void_p)


def build(force: bool = False) -> None:
    """Compile the oracle (and, when the reference tree is mounted, oracle/_ref)."""
    if force or not os.path.exists(_LIB) or (
        os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "grx_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libgrx_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir(os.environ.get("GRX_REFERENCE_ROOT", "/root/reference")):
        # the reference-built artefacts contain THIS engine's kernels (the reference's unchanged
        # client headers and harnesses compiled against include/): stale once a header changed
        import glob
        root = os.path.dirname(_HERE)
        inputs = glob.glob(os.path.join(root, "include", "**", "*.hxx"), recursive=True)
        inputs += [os.path.join(_HERE, f) for f in ("ref_build.sh", "ref_clients_driver.cpp", "ref_driver.cpp")]
        newest = max(os.path.getmtime(f) for f in inputs)
        outputs = [_REF, _REF_CLIENTS, _REF_CLIENTS.replace(".so", "_bucketing.so")]
        outputs += [os.path.join(_HERE, "_ref", "ref_" + a) for a in ("bfs", "sssp", "pr", "kcore", "ppr", "bc", "color", "spmv")]
        if force or any(not os.path.exists(o) or os.path.getmtime(o) < newest for o in outputs):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self) -> None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        self.L = L
        L.orc_mtx_load.restype = C.c_int
        L.orc_mtx_load.argtypes = [C.c_char_p] + [C.POINTER(C.c_int32)] * 3 + [
            C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)),
            C.POINTER(C.POINTER(C.c_float))]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_coo_to_csr.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _f32p, _i32p, _i32p, _f32p]
        L.orc_csr_write_binary.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f32p]
        L.orc_csr_read_binary.argtypes = [C.c_char_p] + [C.POINTER(C.c_int32)] * 3 + [
            C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)),
            C.POINTER(C.POINTER(C.c_float))]
        L.orc_rmat_pair.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64,
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_rmat_weight.restype = C.c_float
        L.orc_rmat_weight.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_rmat_count.restype = C.c_int64
        L.orc_rmat_count.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_int]
        L.orc_rmat_csr.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                   _i32p, _i32p, _f32p]
        L.orc_bfs_heap.restype = C.c_float
        L.orc_bfs_heap.argtypes = [C.c_int32, _i32p, _i32p, C.c_int32, _i32p]
        L.orc_sssp_heap.restype = C.c_float
        L.orc_sssp_heap.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_int32, _f32p]
        L.orc_advance.restype = C.c_int64
        L.orc_advance.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_void_p, C.c_int64,
                                  EDGE_OP, C.c_void_p, C.c_void_p]
        L.orc_filter_bypass.restype = C.c_int64
        L.orc_filter_bypass.argtypes = [_i32p, C.c_int64, VERTEX_OP, C.c_void_p, _i32p]
        L.orc_filter_keep.restype = C.c_int64
        L.orc_filter_keep.argtypes = [_i32p, C.c_int64, VERTEX_OP, C.c_void_p, _i32p]
        L.orc_uniquify.restype = C.c_int64
        L.orc_uniquify.argtypes = [_i32p, C.c_int64, C.c_int]
        L.orc_bfs_frontier.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_int32, _i32p,
                                       C.POINTER(Trace)]
        L.orc_sssp_frontier.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_int32, _f32p,
                                        C.POINTER(Trace)]
        L.orc_pagerank.restype = C.c_int32
        L.orc_pagerank.argtypes = [C.c_int32, _i32p, _i32p, _f32p, C.c_float, C.c_float,
                                   C.c_int32, _f32p]
        L.orc_bfs_levelsync_omp.restype = C.c_float
        L.orc_bfs_levelsync_omp.argtypes = [C.c_int32, _i32p, _i32p, C.c_int32, _i32p,
                                            C.POINTER(C.c_int32)]

    # -- loaders ---------------------------------------------------------
    def mtx_load(self, path: str):
        n, m, nz = C.c_int32(), C.c_int32(), C.c_int32()
        I, J, V = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
        rc = self.L.orc_mtx_load(path.encode(), n, m, nz, I, J, V)
        if rc != 0:
            raise RuntimeError(f"orc_mtx_load({path}) failed: {rc}")
        k = nz.value
        rows = np.ctypeslib.as_array(I, (max(k, 1),))[:k].copy()
        cols = np.ctypeslib.as_array(J, (max(k, 1),))[:k].copy()
        vals = np.ctypeslib.as_array(V, (max(k, 1),))[:k].copy()
        for p in (I, J, V):
            self.L.orc_free(p)
        return n.value, m.value, rows, cols, vals

    def coo_to_csr(self, n_rows, rows, cols, vals):
        nnz = len(rows)
        Ap = np.zeros(n_rows + 1, np.int32)
        Aj = np.zeros(max(nnz, 1), np.int32)
        Ax = np.zeros(max(nnz, 1), np.float32)
        self.L.orc_coo_to_csr(n_rows, nnz, np.ascontiguousarray(rows, np.int32),
                              np.ascontiguousarray(cols, np.int32),
                              np.ascontiguousarray(vals, np.float32), Ap, Aj, Ax)
        return Ap, Aj[:nnz].copy(), Ax[:nnz].copy()

    def mtx_to_csr(self, path: str):
        n, m, rows, cols, vals = self.mtx_load(path)
        return (n,) + self.coo_to_csr(n, rows, cols, vals)

    def csr_write_binary(self, path, n_rows, n_cols, Ap, Aj, Ax):
        self.L.orc_csr_write_binary(path.encode(), n_rows, n_cols, len(Aj), Ap, Aj, Ax)

    def csr_read_binary(self, path):
        n, m, nz = C.c_int32(), C.c_int32(), C.c_int32()
        P, J, V = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
        rc = self.L.orc_csr_read_binary(path.encode(), n, m, nz, P, J, V)
        if rc != 0:
            raise RuntimeError(f"orc_csr_read_binary failed: {rc}")
        Ap = np.ctypeslib.as_array(P, (n.value + 1,)).copy()
        Aj = np.ctypeslib.as_array(J, (max(nz.value, 1),))[:nz.value].copy()
        Ax = np.ctypeslib.as_array(V, (max(nz.value, 1),))[:nz.value].copy()
        for p in (P, J, V):
            self.L.orc_free(p)
        return n.value, m.value, Ap, Aj, Ax

    # -- rmat ------------------------------------------------------------
    def rmat_pair(self, scale, seed, k):
        u, v = C.c_int32(), C.c_int32()
        self.L.orc_rmat_pair(scale, seed, k, u, v)
        return u.value, v.value

    def rmat_csr(self, scale, edge_factor=16, seed=1, weight_seed=0, symmetrize=True):
        n = 1 << scale
        nnz = self.L.orc_rmat_count(scale, edge_factor, seed, int(symmetrize))
        Ap = np.zeros(n + 1, np.int32)
        Aj = np.zeros(max(nnz, 1), np.int32)
        Ax = np.zeros(max(nnz, 1), np.float32)
        self.L.orc_rmat_csr(scale, edge_factor, seed, weight_seed, int(symmetrize), Ap, Aj, Ax)
        return n, Ap, Aj[:nnz], Ax[:nnz]

    # -- checkers --------------------------------------------------------
    def bfs_heap(self, Ap, Aj, source):
        n = len(Ap) - 1
        d = np.empty(n, np.int32)
        ms = self.L.orc_bfs_heap(n, Ap, _nz(Aj, np.int32), source, d)
        return d, ms

    def sssp_heap(self, Ap, Aj, Ax, source):
        n = len(Ap) - 1
        d = np.empty(n, np.float32)
        ms = self.L.orc_sssp_heap(n, Ap, _nz(Aj, np.int32), _nz(Ax, np.float32), source, d)
        return d, ms

    def bfs_levelsync(self, Ap, Aj, source):
        n = len(Ap) - 1
        d = np.empty(n, np.int32)
        th = C.c_int32()
        ms = self.L.orc_bfs_levelsync_omp(n, Ap, _nz(Aj, np.int32), source, d, th)
        return d, ms, th.value

    def bfs_frontier(self, Ap, Aj, Ax, source):
        n = len(Ap) - 1
        d = np.empty(n, np.int32)
        t = Trace()
        self.L.orc_bfs_frontier(n, Ap, _nz(Aj, np.int32), _nz(Ax, np.float32), source, d, t)
        return d, t

    def sssp_frontier(self, Ap, Aj, Ax, source):
        n = len(Ap) - 1
        d = np.empty(n, np.float32)
        t = Trace()
        self.L.orc_sssp_frontier(n, Ap, _nz(Aj, np.int32), _nz(Ax, np.float32), source, d, t)
        return d, t

    def pagerank(self, Ap, Aj, Ax, alpha=0.85, tol=1e-6, max_iter=0):
        n = len(Ap) - 1
        p = np.empty(n, np.float32)
        it = self.L.orc_pagerank(n, Ap, _nz(Aj, np.int32), _nz(Ax, np.float32), alpha, tol,
                                 max_iter, p)
        return p, it

    # -- operators -------------------------------------------------------
    def advance(self, Ap, Aj, Ax, frontier, op, want_output=True):
        """op(src, dst, edge, w) -> bool, a Python callable (small inputs only)."""
        n = len(Ap) - 1
        cb = EDGE_OP(lambda s, d, e, w, _c: int(bool(op(s, d, e, w))))
        if frontier is None:
            inp, n_in = None, n
            total = int(Ap[n])
        else:
            f = np.ascontiguousarray(frontier, np.int32)
            inp, n_in = f.ctypes.data_as(C.c_void_p), len(f)
            valid = f[f != -1]
            total = int((Ap[valid + 1] - Ap[valid]).sum()) if len(valid) else 0
        out = np.empty(max(total, 1), np.int32)
        got = self.L.orc_advance(n, Ap, _nz(Aj, np.int32), _nz(Ax, np.float32), inp, n_in, cb, None,
                                 out.ctypes.data_as(C.c_void_p) if want_output else None)
        assert got == total
        return out[:total]

    def filter_bypass(self, frontier, pred):
        f = np.ascontiguousarray(frontier, np.int32)
        out = np.empty(max(len(f), 1), np.int32)
        cb = VERTEX_OP(lambda v, _c: int(bool(pred(v))))
        self.L.orc_filter_bypass(_nz(f, np.int32), len(f), cb, None, out)
        return out[:len(f)]

    def filter_keep(self, frontier, pred):
        f = np.ascontiguousarray(frontier, np.int32)
        out = np.empty(max(len(f), 1), np.int32)
        cb = VERTEX_OP(lambda v, _c: int(bool(pred(v))))
        m = self.L.orc_filter_keep(_nz(f, np.int32), len(f), cb, None, out)
        return out[:m].copy()

    def uniquify(self, frontier, sort=True):
        f = np.array(frontier, np.int32, copy=True)
        if len(f) == 0:
            return f
        m = self.L.orc_uniquify(f, len(f), int(sort))
        return f[:m].copy()


def _nz(a, dt):
    """ndpointer rejects zero-length views of some shapes; hand C a 1-element dummy."""
    a = np.ascontiguousarray(a, dt)
    return a if a.size else np.zeros(1, dt)


class RefOracle:
    """The reference's own CPU checkers (compiled in place by oracle/ref_build.sh)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(_REF)

    def __init__(self) -> None:
        L = C.CDLL(_REF)
        self.L = L
        L.ref_bfs_cpu.restype = C.c_float
        L.ref_bfs_cpu.argtypes = [C.c_int, _i32p, _i32p, C.c_int, _i32p]
        L.ref_sssp_cpu.restype = C.c_float
        L.ref_sssp_cpu.argtypes = [C.c_int, _i32p, _i32p, _f32p, C.c_int, _f32p]

    def bfs(self, Ap, Aj, source):
        n = len(Ap) - 1
        d = np.empty(n, np.int32)
        ms = self.L.ref_bfs_cpu(n, Ap, _nz(Aj, np.int32), source, d)
        return d, ms

    def sssp(self, Ap, Aj, Ax, source):
        n = len(Ap) - 1
        d = np.empty(n, np.float32)
        ms = self.L.ref_sssp_cpu(n, Ap, _nz(Aj, np.int32), _nz(Ax, np.float32), source, d)
        return d, ms


class RefClients:
    """The reference's UNCHANGED bfs.hxx / sssp.hxx / pr.hxx compiled against this repository's
    include/gunrock (oracle/ref_clients_driver.cpp -> oracle/_ref/libgrx_ref_clients.so).
    Drop-in evidence for the GPU tests; takes torch device tensors."""

    @staticmethod
    def available(variant: str = "") -> bool:
        return os.path.exists(RefClients._path(variant))

    @staticmethod
    def _path(variant: str = "") -> str:
        return _REF_CLIENTS if not variant else _REF_CLIENTS.replace(".so", f"_{variant}.so")

    def __init__(self, variant: str = "") -> None:
        import torch  # noqa: F401  (one HIP runtime per process: torch's first)
        L = C.CDLL(self._path(variant))
        self.L = L
        vp = C.c_void_p
        L.refc_bfs.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.POINTER(C.c_float)]
        L.refc_sssp.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.POINTER(C.c_float)]
        L.refc_pr.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_float, C.c_float, vp,
                              C.POINTER(C.c_float)]

        job = [C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.POINTER(C.c_float), C.c_int, C.c_int,
               C.c_int, C.c_int, C.c_int, vp, vp, vp]
        if hasattr(L, "refc_bfs_job"):
            L.refc_bfs_job.argtypes = job
            L.refc_sssp_job.argtypes = job
            L.refc_pr_job_refused.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp]

    def tc(self, ap, aj, ax, counts):
        """tc::run(G, reduce_all=True, counts, &total) -> total triangles."""
        ms, total = C.c_float(), C.c_ulonglong()
        self.L.refc_tc.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.POINTER(C.c_ulonglong),
                                                                           C.POINTER(C.c_float)]
        rc = self.L.refc_tc(ap.numel() - 1, aj.numel(), ap.data_ptr(), aj.data_ptr(), ax.data_ptr(),
                            counts.data_ptr(), total, ms)
        assert rc == 0
        return total.value

    def run_job(self, algo, ap, aj, ax, source, out, rank, world, lo, hi, unique_id=None,
                all_gather=None, all_reduce=None):
        """bfs.hxx / sssp.hxx unchanged as one rank of a vertex-partitioned job.  ap/aj/ax = the
        rank's slice.  unique_id: 128 bytes -> RCCL; else all_gather / all_reduce = ctypes callbacks
        (essentials_amd.api.ALL_GATHER_FN / ALL_REDUCE_FN)."""
        ms = C.c_float()
        fn = self.L.refc_bfs_job if algo == "bfs" else self.L.refc_sssp_job
        idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        rc = fn(ap.numel() - 1, aj.numel(), ap.data_ptr(), aj.data_ptr(), ax.data_ptr(), source,
                out.data_ptr(), ms, rank, world, lo, hi, 1 if unique_id is not None else 0, idbuf,
                C.cast(all_gather, C.c_void_p) if all_gather else None,
                C.cast(all_reduce, C.c_void_p) if all_reduce else None)
        assert rc == 0, f"{algo} job failed"
        return ms.value

    def pr_job_refused(self, ap, aj, ax, p, unique_id) -> bool:
        idbuf = C.create_string_buffer(unique_id, 128)
        return self.L.refc_pr_job_refused(ap.numel() - 1, aj.numel(), ap.data_ptr(), aj.data_ptr(),
                                          ax.data_ptr(), p.data_ptr(), idbuf) == 1

    def bfs(self, ap, aj, ax, source, out):
        ms = C.c_float()
        rc = self.L.refc_bfs(ap.numel() - 1, aj.numel(), ap.data_ptr(), aj.data_ptr(), ax.data_ptr(),
                             source, out.data_ptr(), ms)
        assert rc == 0
        return ms.value

    def sssp(self, ap, aj, ax, source, out):
        ms = C.c_float()
        rc = self.L.refc_sssp(ap.numel() - 1, aj.numel(), ap.data_ptr(), aj.data_ptr(),
                              ax.data_ptr(), source, out.data_ptr(), ms)
        assert rc == 0
        return ms.value

    def pr(self, ap, aj, ax, alpha, tol, out):
        ms = C.c_float()
        rc = self.L.refc_pr(ap.numel() - 1, aj.numel(), ap.data_ptr(), aj.data_ptr(), ax.data_ptr(),
                            alpha, tol, out.data_ptr(), ms)
        assert rc == 0
        return ms.value
