"""Runners used by bench.py and the multi-process tests: one graph resident in HBM, repeated
BFS / SSSP traversals through the C ABI.

SingleRunner          one GPU, the whole graph (grx_bfs / grx_sssp).
PartitionedTraversal  the multi-GPU superstep protocol: one process per GPU,
                      1-D edge-balanced vertex partition, replicated labels, local advance,
                      ALL-GATHER of the per-rank output frontiers between supersteps
                      (torch.distributed: backend "nccl" is RCCL over xGMI), min-combine +
                      owner admission.  The local kernels come from a `kernels` object:
                      HipKernels (the C ABI, production) -- tests may pass another object with
                      the same methods to exercise the protocol on CPU ranks over gloo.
PartitionedPageRank   PageRank on the same partition: local scatter + all-reduce of the partials.
PartitionedRunner     bench-side wrapper: R-MAT graph, slicing, repeated traversals.

Exchange format (one int64 slot per rank): word 0 = number of (vertex,label) pairs the rank
found this superstep, words 1.. = pairs (low 32 bits vertex id, high 32 bits label bits).
A first all-gather moves a SMALL fixed slot (header + the first pairs); only when some rank found
more than fits is a second all-gather issued with a slot sized by the largest count, which every
rank knows from the first -- so most supersteps cost one collective and no count exchange.
Dense supersteps use a smaller message instead of the big pair slot: BFS all-gathers per-rank LEVEL
BITMAPS (V/8 bytes, labels implied), SSSP ALL-REDUCES (MIN) the distance replicas in place and the
next step admits what fell below its pre-advance snapshot (grx_recv_format in essentials_amd.h).
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import api as ea

HBM_PEAK_GBPS = 8000.0
OP_BFS, OP_SSSP = int(ea.EdgeOp.bfs), int(ea.EdgeOp.sssp)
RECV_PAIRS, RECV_LEVEL_BITMAP, RECV_REPLICA_MIN = 0, 1, 2   # grx_recv_format


def bfs_algorithmic_bytes(edges_traversed: int, vertices_reached: int) -> int:
    """per traversed edge 4 B column index + 4 B label gather; per reached vertex
    4 B frontier read + 8 B row offsets + 4 B label write + 4 B next-frontier write."""
    return 8 * edges_traversed + 20 * vertices_reached


def partition_bounds(row_offsets, world: int):
    """Edge-balanced 1-D split points (the rule grx_graph_partition applies): rank k owns rows
    [b[k], b[k+1]) where b[k] is the first row whose offset reaches k*E/world."""
    ap = np.asarray(row_offsets)
    n = len(ap) - 1
    e = int(ap[n])
    b = [0]
    for k in range(1, world):
        b.append(int(np.searchsorted(ap, np.int64(e) * k // world, side="left")))
    b.append(n)
    for k in range(1, len(b)):
        b[k] = min(max(b[k], b[k - 1]), n)
    return b


# --------------------------------------------------------------------------- single GPU
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
        self.runs = {"bfs": 0, "bfs_call_every_edge": 0, "sssp": 0, "sssp_two_pass": 0,
                     "bfs_direction_optimized": 0}
        self.enact_ms = {"bfs": [], "sssp": []}   # per timed call (reset by flush_edges)

    def host_csr(self):
        if self._host is None:
            self._host = self.g.to_host()
        return self._host

    def global_degrees(self):
        return np.diff(self.g.offsets_to_host()).astype(np.int64)

    def bfs(self, source: int, opts: ea.Options) -> int:
        _, st = ea.bfs(self.ctx, self.g, source, self.depth, opts)
        self.last["bfs"] = st
        self.runs["bfs"] += 1
        self.enact_ms["bfs"].append(st.elapsed_ms)
        return st.edges_traversed

    def sssp(self, source: int, opts: ea.Options) -> int:
        _, st = ea.sssp(self.ctx, self.g, source, self.dist, opts)
        self.last["sssp"] = st
        self.runs["sssp"] += 1
        self.enact_ms["sssp"].append(st.elapsed_ms)
        return st.edges_traversed

    def flush_edges(self) -> int:
        return 0   # bfs() / sssp() already returned their counts

    def mark(self) -> None:
        """Start a new window of per-call enact times (bench: the timed steps)."""
        self.enact_ms = {"bfs": [], "sssp": []}

    def sssp_roofline(self, source: int, lb) -> dict:
        """SSSP advance = 12 B per relaxation executed (column index, weight, distance
        gather) + 20 B per valid frontier entry; the reference's two-pass form adds 8 B per entry
        (bypass filter: read + write).  Kernel time = HIP events around every advance launch."""
        best = None
        for _ in range(3):
            _, st = ea.sssp(self.ctx, self.g, source, self.dist,
                            ea.Options(load_balance=lb, collect_kernel_time=True))
            self.runs["sssp"] += 1
            if best is None or st.advance_kernel_ms < best.advance_kernel_ms:
                best = st
        relax, slots = best.edges_expanded, sum(best.frontier_slots)
        nbytes = 12 * relax + 20 * slots
        achieved = nbytes / (best.advance_kernel_ms * 1e-3) / 1e9
        two = None
        for _ in range(3):
            _, st = ea.sssp(self.ctx, self.g, source, self.dist,
                            ea.Options(load_balance=lb, sssp_two_pass=True))
            self.runs["sssp_two_pass"] += 1
            if two is None or st.elapsed_ms < two.elapsed_ms:
                two = st
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "block_mapped advance kernels of one SSSP (relax functor, all iterations)",
                "algorithmic_bytes": nbytes, "relaxations": relax, "frontier_entries": slots,
                "kernel_ms": best.advance_kernel_ms, "enact_ms": best.elapsed_ms,
                "iterations": best.iterations, "source": source,
                "relaxations_per_second_g": relax / (best.advance_kernel_ms * 1e-3) / 1e9,
                "two_pass_reference_formulation": {
                    "enact_ms": two.elapsed_ms, "iterations": two.iterations,
                    "relaxations_upper_bound": two.edges_expanded,
                    "frontier_entries": sum(two.frontier_slots),
                    "relaxations_note": "sum of the input frontiers' work hints; the bypass filter turns "
                                        "duplicates into holes and passes the advance's hint on, so "
                                        "this counts the degrees of removed duplicates too",
                    "note": "advance + bypass filter with the racy stamp test, as reference "
                            "algorithms/sssp.hxx:110-144 (grx_options.sssp_two_pass); same distances"}}

    def bfs_roofline(self, source: int, lb) -> dict:
        """Kernel-level roofline of one BFS: HIP events around every advance launch."""
        best = None
        for _ in range(3):
            _, st = ea.bfs(self.ctx, self.g, source, self.depth,
                           ea.Options(load_balance=lb, collect_kernel_time=True))
            self.runs["bfs"] += 1
            if best is None or st.advance_kernel_ms < best.advance_kernel_ms:
                best = st
        nbytes = bfs_algorithmic_bytes(best.edges_traversed, best.vertices_reached)
        achieved = nbytes / (best.advance_kernel_ms * 1e-3) / 1e9
        every = None   # the same search with the functor called for EVERY edge (no settled hint)
        for _ in range(3):
            _, st = ea.bfs(self.ctx, self.g, source, self.depth,
                           ea.Options(load_balance=lb, collect_kernel_time=True, call_every_edge=True))
            self.runs["bfs_call_every_edge"] += 1
            if every is None or st.advance_kernel_ms < every.advance_kernel_ms:
                every = st
        # the ceiling this functor actually sits under: one random 4-B label lookup per edge
        # (grx_measure_gather_rate: table[column[i]] over the graph's own column array)
        gather = {m: self.ctx.gather_rate(self.g, mode) / 1e9 for m, mode in (("agent_scope", 1), ("plain", 0))}
        kernel_gteps = best.edges_traversed / (best.advance_kernel_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "gather_roof": {"unit": "G lookups/s", "agent_scope_loads": gather["agent_scope"],
                                "plain_loads": gather["plain"],
                                "frac_of_gather_roof": kernel_gteps / max(gather.values()),
                                "note": "measured live: acc += table[column[i]] over all edges of this "
                                        "graph (4-B entries, |V| of them), no frontier logic, atomics "
                                        "or output -- the ceiling of any label-testing advance"},
                "kernel": "BFS advance, all levels of one traversal: settled-bitmap rebuild + "
                          "classify_hubs_kernel + expand_settled_kernel on the wide levels, "
                          "block_mapped_kernel + chunk_kernel on the others",
                "call_every_edge_formulation": {
                    "kernel_ms": every.advance_kernel_ms, "enact_ms": every.elapsed_ms,
                    "frac": nbytes / (every.advance_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "note": "grx_options.call_every_edge: no settled-destination hint, the functor (label "
                            "test + atomic min) runs for every edge -- all the engine can do for the "
                            "unchanged bfs.hxx; same depths"},
                "algorithmic_bytes": nbytes, "kernel_ms": best.advance_kernel_ms,
                "launches": best.advance_launches, "enact_ms": best.elapsed_ms,
                "edges_traversed": best.edges_traversed, "vertices_reached": best.vertices_reached,
                "kernel_gteps": best.edges_traversed / (best.advance_kernel_ms * 1e-3) / 1e9}

    def reference_clients(self, sources, lb) -> dict:
        """What the reference's UNCHANGED bfs.hxx + sssp.hxx get from this engine, over the same
        sources as the timed steps (outside the timed region): the BFS whose functor runs for every
        edge (no settled hint to give; grx_options.call_every_edge) and the two-pass SSSP (advance +
        bypass filter with the racy stamp test, algorithms/sssp.hxx:110-144; grx_options.sssp_two_pass),
        both on the caller's vertex numbering (no hot-first copy: an unchanged client traverses the
        graph_t it was given).  tests/test_gpu_reference_clients.py runs the real headers, compiled
        in place, through the same engine; these two options are their formulation behind the C ABI."""
        b_ms, s_ms, edges = [], [], 0
        ob = ea.Options(load_balance=lb, call_every_edge=True)
        os_ = ea.Options(load_balance=lb, sssp_two_pass=True)
        for s in sources:
            _, st = ea.bfs(self.ctx, self.g, s, self.depth, ob)
            self.runs["bfs_call_every_edge"] += 1
            b_ms.append(st.elapsed_ms)
            edges += st.edges_traversed
            _, st = ea.sssp(self.ctx, self.g, s, self.dist, os_)
            self.runs["sssp_two_pass"] += 1
            s_ms.append(st.elapsed_ms)
            edges += st.edges_traversed
        step = (sum(b_ms) + sum(s_ms)) / len(sources)
        return {"bfs_every_edge_enact_ms": sum(b_ms) / len(b_ms),
                "sssp_two_pass_enact_ms": sum(s_ms) / len(s_ms),
                "step_ms": step, "mteps": edges / ((sum(b_ms) + sum(s_ms)) * 1e-3) / 1e6,
                "sources": len(sources),
                "note": "enact() time only (the headline's ms_per_step also holds ~0.2 ms per step outside "
                        "enact()); caller's numbering, no settled hint, two-pass SSSP"}

    def bfs_direction_optimized(self, sources, lb) -> dict:
        """SURVEY 8(f) rank 4 (beyond the reference, whose advance throws for pull): the same
        traversals with the wide levels pulled.  Same depths; reported beside the headline."""
        ms, edges, pulls = [], 0, 0
        opts = ea.Options(load_balance=lb, direction_optimized=True)
        for s in sources:
            best = None
            for _ in range(2):
                _, st = ea.bfs(self.ctx, self.g, s, self.depth, opts)
                self.runs["bfs_direction_optimized"] += 1
                if best is None or st.elapsed_ms < best.elapsed_ms:
                    best = st
            ms.append(best.elapsed_ms)
            edges += best.edges_traversed
            pulls += best.pull_iterations
        total = sum(ms)
        nbytes = 8 * edges + 20 * best.vertices_reached * len(sources)
        return {"enact_ms_mean": total / len(ms), "mteps": edges / total / 1e3,
                "pull_levels_mean": pulls / len(ms), "sources": len(sources),
                "effective_algorithmic_gbps": nbytes / (total * 1e-3) / 1e9,
                "note": "push levels: block_mapped; wide levels: pull (advance_direction_t::backward); "
                        "edges counted as for the push search (out-degrees of reached vertices)"}

    def detail(self) -> dict:
        d = {}
        for k, st in self.last.items():
            d[k] = {"enact_ms": st.elapsed_ms, "iterations": st.iterations,
                    "enact_ms_mean_over_timed_steps":
                        (sum(self.enact_ms[k]) / len(self.enact_ms[k])) if self.enact_ms.get(k) else None,
                    "edges_traversed": st.edges_traversed, "vertices_reached": st.vertices_reached,
                    "mteps_enact": st.edges_traversed / max(st.elapsed_ms, 1e-9) / 1e3,
                    "edges_expanded": st.edges_expanded,
                    "mteps_expanded": st.edges_expanded / max(st.elapsed_ms, 1e-9) / 1e3,
                    "frontier_slots": st.frontier_slots[:16]}
        return d


# --------------------------------------------------------------------------- multi GPU
class HostStagedCollectives:
    """The two collectives of a job as HOST callbacks over a torch.distributed process group whose
    backend moves host memory (gloo): device -> host (grx_copy_to_host), the collective on CPU
    tensors, host -> device.  What the test rigs attach with Context.attach_collectives so that the
    C++ superstep loop (grx_partitioned_run) runs with several ranks sharing one GPU; production
    attaches RCCL instead (attach_job)."""

    _NP = {0: np.int32, 1: np.float32, 2: np.int64}

    def __init__(self, ctx: ea.Context, dist):
        import torch
        self.torch, self.ctx, self.dist = torch, ctx, dist
        self.world = dist.get_world_size()
        self.calls = {"all_gather": 0, "all_reduce": 0}

    def all_gather(self, d_send: int, d_recv: int, nbytes: int, stream: int) -> int:
        torch = self.torch
        self.calls["all_gather"] += 1
        mine = np.empty(nbytes, np.uint8)
        self.ctx.copy_to_host(mine, d_send)          # drains the engine's stream first
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, torch.from_numpy(mine))
        self.ctx.copy_to_device(d_recv, torch.cat(parts).numpy())
        return 0

    def all_reduce(self, d_buffer: int, count: int, dtype: int, op: int, stream: int) -> int:
        torch = self.torch
        self.calls["all_reduce"] += 1
        h = np.empty(count, self._NP[dtype])
        self.ctx.copy_to_host(h, d_buffer)
        t = torch.from_numpy(h)
        rop = {0: self.dist.ReduceOp.MIN, 1: self.dist.ReduceOp.SUM, 2: self.dist.ReduceOp.MAX}[op]
        self.dist.all_reduce(t, op=rop)
        self.ctx.copy_to_device(d_buffer, h)
        return 0


def attach_job(ctx: ea.Context, dist, transport: str | None = None) -> str:
    """Attach `ctx` to the job `dist` (an initialised torch.distributed) describes.
    transport "rccl": the engine's OWN RCCL communicator (ncclCommInitRank; rank 0's unique id
    travels through the process group) -- the C++ loop then calls ncclAllGather / ncclAllReduce
    itself; "hooks": host-staged callbacks over the process group.  Default: rccl when the group's
    backend is nccl, hooks otherwise.  Returns the transport attached."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if transport is None:
        transport = "rccl" if dist.get_backend() == "nccl" else "hooks"
    if transport == "rccl":
        box = [ea.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.attach_rccl(rank, world, box[0])
    elif transport == "hooks":
        ctx._collectives = HostStagedCollectives(ctx, dist)
        ctx.attach_collectives(rank, world, ctx._collectives.all_gather, ctx._collectives.all_reduce)
    else:
        raise ValueError(f"unknown transport {transport!r}")
    return transport


class HipKernels:
    """The production local kernels: grx_partitioned_expand / grx_partitioned_admit."""

    def __init__(self, ctx: ea.Context, local_graph: ea.Graph, options: ea.Options | None = None):
        self.ctx, self.g = ctx, local_graph
        self.opts = (options or ea.Options())._c()
        self.lib = ea.load_library()

    def expand(self, op, labels, rnd, frontier, n_frontier, scratch, sent, send) -> int:
        n = C.c_int64()
        self.ctx.after_torch()
        ea._check(self.lib.grx_partitioned_expand(
            self.ctx._h, self.g._h, C.byref(self.opts), op, labels.data_ptr(), rnd,
            frontier.data_ptr() if n_frontier else None, n_frontier, scratch.data_ptr(),
            scratch.numel(), sent.data_ptr(), send.data_ptr(), send.numel(), C.byref(n)),
            "grx_partitioned_expand")
        return n.value

    def admit(self, op, labels, stamp, rnd, recv, fmt, world, slot, rank, lo, hi, nxt):
        n_next = C.c_int64()
        self.ctx.after_torch()
        ea._check(self.lib.grx_partitioned_admit(
            self.ctx._h, op, labels.data_ptr(), labels.numel(), stamp.data_ptr(), rnd,
            recv.data_ptr(), fmt, world, slot, rank, lo, hi, nxt.data_ptr(), nxt.numel(),
            C.byref(n_next)), "grx_partitioned_admit")
        return n_next.value

    def level_bitmap(self, depth, level, words) -> None:
        """words[ceil(V/64)] <- bit v = (depth[v] == level); enqueue-only."""
        self.ctx.after_torch()
        ea._check(self.lib.grx_partitioned_level_bitmap(
            self.ctx._h, depth.data_ptr(), depth.numel(), level, words.data_ptr(), words.numel()),
            "grx_partitioned_level_bitmap")

    def pr_scatter(self, alpha, p, scale, compute_scale, partial, lo, hi) -> None:
        """grx_pagerank_partitioned_scatter: partial <- this rank's PageRank contributions."""
        self.ctx.after_torch()
        ea._check(self.lib.grx_pagerank_partitioned_scatter(
            self.ctx._h, self.g._h, alpha, p.data_ptr(), scale.data_ptr(), int(compute_scale),
            partial.data_ptr(), lo, hi, C.byref(self.opts)), "grx_pagerank_partitioned_scatter")

    def step(self, op, labels, stamp, sent, rnd, recv, fmt, world, slot, rank, lo, hi, frontier,
             fcount, scratch, send, snapshot=None) -> None:
        """Fused, enqueue-only superstep (grx_partitioned_step): admit `recv` -> [snapshot the owned
        labels ->] advance -> pack."""
        self.ctx.after_torch()
        ea._check(self.lib.grx_partitioned_step(
            self.ctx._h, self.g._h, C.byref(self.opts), op, labels.data_ptr(), stamp.data_ptr(),
            sent.data_ptr(), rnd, recv.data_ptr() if recv is not None else None, fmt, world, slot,
            rank, lo, hi, frontier.data_ptr(), frontier.numel(), fcount.data_ptr(),
            scratch.data_ptr(), scratch.numel(), send.data_ptr(), send.numel(),
            snapshot.data_ptr() if snapshot is not None else None), "grx_partitioned_step")


class PartitionedPageRank:
    """PageRank on the vertex partition: every rank keeps a replica of p, scatters
    from the rows it owns into a private partial vector (`kernels.pr_scatter`, production:
    grx_pagerank_partitioned_scatter), the partials are ALL-REDUCED (SUM; RCCL on `nccl`) and every
    rank applies the same update, so the replicas stay identical and the stop test
    (max |p - p_previous| < tol after >= 1 iteration, pr.hxx:155-178) needs no extra collective."""

    def __init__(self, kernels, dist, rank: int, world: int, n_global: int, lo: int, hi: int, device,
                 stream=None):
        """stream: the torch stream the engine context was created on (so that the scatter kernels,
        torch's update arithmetic and the all-reduce are ordered by ONE stream); None: the wrappers
        order the engine's stream after torch's before every call (HipKernels -> after_torch)."""
        import torch
        self.torch, self.k, self.dist = torch, kernels, dist
        self.stream = stream
        self.rank, self.world, self.n, self.lo, self.hi = rank, world, n_global, lo, hi
        f32 = torch.float32
        self.scale = torch.zeros(n_global, dtype=f32, device=device)
        self.partial = torch.zeros(n_global + 1, dtype=f32, device=device)
        self._backend = dist.get_backend() if dist is not None and world > 1 else None

    def _all_reduce(self, t):
        if self.world == 1:
            return
        if self._backend == "nccl" or not t.is_cuda:
            self.dist.all_reduce(t)
        else:  # gloo with device tensors (test rigs): stage through the host
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)

    def run(self, p, alpha: float = 0.85, tol: float = 1e-6, max_iterations: int = 0) -> dict:
        """p: replica [V] float32, overwritten with the ranks."""
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):
                return self._run(p, alpha, tol, max_iterations)
        return self._run(p, alpha, tol, max_iterations)

    def _run(self, p, alpha, tol, max_iterations) -> dict:
        torch = self.torch
        p.fill_(1.0 / self.n)
        if p.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 0
        while True:
            self.k.pr_scatter(alpha, p, self.scale, it == 0, self.partial, self.lo, self.hi)
            self._all_reduce(self.partial)
            new = self.partial[: self.n] + (1.0 - alpha + self.partial[self.n]) / self.n
            err = float((new - p).abs().max())
            p.copy_(new)
            it += 1
            if err < tol or (max_iterations and it >= max_iterations):
                break
        if p.is_cuda:
            torch.cuda.synchronize()
        return {"elapsed_ms": (time.perf_counter() - t0) * 1e3, "iterations": it, "last_error": err}


class PartitionedTraversal:
    """BSP supersteps of a vertex-partitioned BFS / SSSP over `dist` (torch.distributed or None)."""

    SMALL_SLOT = 1 << 15  # int64 words per rank in the first all-gather (256 KiB)

    def __init__(self, kernels, dist, rank: int, world: int, n_global: int, lo: int, hi: int,
                 local_nnz: int, device, small_slot: int | None = None, fused: bool = True,
                 stream=None, dense_threshold: int | None = None,
                 replica_threshold: int | None = None):
        """fused=True uses kernels.step (one enqueue-only call + one host synchronisation per
        superstep); it needs the engine context and the collectives on ONE stream: pass that
        torch stream as `stream` (the context must have been created on stream.cuda_stream)."""
        import torch
        self.torch = torch
        # without a shared stream the fused loop would race on device tensors: use the two-call loop
        on_device = str(device).startswith("cuda")
        self.fused = fused and hasattr(kernels, "step") and (stream is not None or not on_device)
        self.stream = stream
        self.k, self.dist = kernels, dist
        self.rank, self.world, self.n, self.lo, self.hi = rank, world, n_global, lo, hi
        self.device = device
        # every rank derives the same slot sizes from (small_slot, V)
        self.slot0 = max(2, min(int(small_slot or self.SMALL_SLOT), n_global + 2))
        i32, i64 = torch.int32, torch.int64
        self.stamp = torch.full((n_global,), -1, dtype=i32, device=device)
        self.sent = torch.full((n_global,), -1, dtype=i32, device=device)
        own = max(hi - lo, 1)
        self.frontier = [torch.empty(own + 64, dtype=i32, device=device) for _ in range(2)]
        self.scratch = torch.empty(max(local_nnz, 1) + n_global + 64, dtype=i32, device=device)
        self.send = torch.zeros(n_global + 2, dtype=i64, device=device)
        self.fcount = torch.zeros(1, dtype=i64, device=device)
        self.recv = torch.zeros(world * self.slot0, dtype=i64, device=device)
        self._recv_big = None
        # dense BFS supersteps exchange level bitmaps (V/8 bytes per rank) instead of pairs (8 bytes
        # per discovery): from V/64 discoveries on some rank the bitmap is the smaller message
        self.words = (n_global + 63) // 64
        self.dense_threshold = int(dense_threshold if dense_threshold is not None
                                   else max(n_global // 64, self.slot0 - 1))
        self.can_dense = hasattr(kernels, "level_bitmap")
        # dense SSSP supersteps all-reduce (MIN) the distance replicas instead of gathering pairs:
        # 2 x 4 V bytes through the ring against world x 8 bytes per find of the busiest rank
        self.replica_threshold = int(replica_threshold if replica_threshold is not None
                                     else max(n_global // max(world, 1), self.slot0 - 1))
        self.snapshot = torch.zeros(n_global, dtype=torch.float32, device=device)
        self.bits = torch.zeros(self.words, dtype=i64, device=device)
        self.recv_bits = torch.zeros(world * self.words, dtype=i64, device=device)
        self._backend = dist.get_backend() if dist is not None and world > 1 else None

    # -- the collective -------------------------------------------------------------------
    def _all_gather(self, recv, send):
        """recv[world*slot] <- concatenation of every rank's send[slot]."""
        if self.world == 1:
            recv[: send.numel()].copy_(send)
            return
        if self._backend == "nccl" or not send.is_cuda:
            self.dist.all_gather_into_tensor(recv, send)
        else:  # gloo with device tensors (test rigs): stage through the host
            h = send.cpu()
            parts = [self.torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(parts, h)
            recv.copy_(self.torch.cat(parts).to(recv.device))

    def _all_reduce_min(self, labels):
        if self.world == 1:
            return
        op = self.dist.ReduceOp.MIN
        if self._backend == "nccl" or not labels.is_cuda:
            self.dist.all_reduce(labels, op=op)
        else:  # gloo with device tensors (test rigs): stage through the host
            h = labels.cpu()
            self.dist.all_reduce(h, op=op)
            labels.copy_(h)

    def run(self, op: int, source: int, labels) -> dict:
        """labels: replica [V] (int32 for BFS, float32 for SSSP), overwritten."""
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):
                return self._run_fused(op, source, labels) if self.fused else self._run(op, source, labels)
        return self._run_fused(op, source, labels) if self.fused else self._run(op, source, labels)

    def _run_fused(self, op: int, source: int, labels) -> dict:
        torch = self.torch
        unreached = ea.INT_UNREACHED if op == OP_BFS else ea.FLT_UNREACHED
        labels.fill_(unreached)
        labels[source] = 0
        self.stamp.fill_(-1)
        self.sent.fill_(-1)
        frontier = self.frontier[0]
        owned = self.lo <= source < self.hi
        if owned:
            frontier[0] = source
        self.fcount.fill_(1 if owned else 0)
        if labels.is_cuda:
            torch.cuda.current_stream().synchronize()
        t0 = time.perf_counter()
        rounds = found_total = collectives = dense = reduced = 0
        recv_prev, slot_prev, fmt_prev = None, 0, RECV_PAIRS
        snapshot = self.snapshot if op == OP_SSSP else None
        while True:
            self.k.step(op, labels, self.stamp, self.sent, rounds, recv_prev, fmt_prev, self.world,
                        slot_prev, self.rank, self.lo, self.hi, frontier, self.fcount, self.scratch,
                        self.send, snapshot)
            slot, fmt = self.slot0, RECV_PAIRS
            recv = self.recv
            self._all_gather(recv, self.send[:slot])
            collectives += 1
            heads = recv.view(self.world, slot)[:, 0]
            counts = heads.cpu() if heads.is_cuda else heads.clone()   # the one host wait
            most = int(counts.max())
            if most == 0:
                break
            if op == OP_BFS and self.can_dense and most > self.dense_threshold:
                # every rank takes this branch together (same gathered counts)
                self.k.level_bitmap(labels, rounds + 1, self.bits)
                recv, slot, fmt = self.recv_bits, self.words, RECV_LEVEL_BITMAP
                self._all_gather(recv, self.bits)      # same stream: ordered after the bitmap kernel
                collectives += 1
                dense += 1
            elif op == OP_SSSP and most > self.replica_threshold:
                # the replicas already hold every rank's own improvements: combine them in place;
                # the next step admits what fell below its snapshot
                self._all_reduce_min(labels)
                recv, slot, fmt = self.snapshot, 0, RECV_REPLICA_MIN
                collectives += 1
                reduced += 1
            elif most > slot - 1:
                slot = min(((most + 1 + 4095) // 4096) * 4096, self.send.numel())
                if self._recv_big is None or self._recv_big.numel() < self.world * slot:
                    self._recv_big = torch.zeros(self.world * slot, dtype=torch.int64,
                                                 device=self.device)
                recv = self._recv_big[: self.world * slot]
                self._all_gather(recv, self.send[:slot])   # same stream: ordered before the admit
                collectives += 1
            found_total += int(counts.sum())
            recv_prev, slot_prev, fmt_prev = recv, slot, fmt
            rounds += 1
        if labels.is_cuda:
            torch.cuda.current_stream().synchronize()
        return {"elapsed_ms": (time.perf_counter() - t0) * 1e3, "supersteps": rounds + 1,
                "pairs_exchanged": found_total, "collectives": collectives, "fused": True,
                "bitmap_supersteps": dense, "allreduce_supersteps": reduced}

    def _run(self, op: int, source: int, labels) -> dict:
        torch = self.torch
        unreached = ea.INT_UNREACHED if op == OP_BFS else ea.FLT_UNREACHED
        labels.fill_(unreached)
        labels[source] = 0
        self.stamp.fill_(-1)
        self.sent.fill_(-1)
        cur, nxt = self.frontier
        n_cur = 0
        if self.lo <= source < self.hi:
            cur[0] = source
            n_cur = 1
        if labels.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        rounds = found_total = collectives = dense = 0
        prof = [0.0, 0.0, 0.0, 0.0] if os.environ.get("GRX_PART_PROFILE") else None
        while True:
            ta = time.perf_counter()
            self.k.expand(op, labels, rounds, cur, n_cur, self.scratch, self.sent, self.send)
            tb = time.perf_counter()
            slot = self.slot0
            recv = self.recv
            self._all_gather(recv, self.send[:slot])
            collectives += 1
            heads = recv.view(self.world, slot)[:, 0]
            counts = heads.cpu() if heads.is_cuda else heads
            most = int(counts.max())
            tc = time.perf_counter()
            if prof is not None:
                prof[0] += tb - ta
                prof[1] += tc - tb
            if most == 0:
                break  # no rank improved anything: every replica is final
            fmt = RECV_PAIRS
            if op == OP_BFS and self.can_dense and most > self.dense_threshold:
                self.k.level_bitmap(labels, rounds + 1, self.bits)
                if recv.is_cuda:
                    self.k.ctx.synchronize()       # the engine's stream wrote self.bits
                recv, slot, fmt = self.recv_bits, self.words, RECV_LEVEL_BITMAP
                self._all_gather(recv, self.bits)
                collectives += 1
                dense += 1
                if recv.is_cuda:
                    torch.cuda.current_stream().synchronize()
            elif most > slot - 1:
                slot = min(((most + 1 + 4095) // 4096) * 4096, self.send.numel())
                if self._recv_big is None or self._recv_big.numel() < self.world * slot:
                    self._recv_big = torch.zeros(self.world * slot, dtype=torch.int64,
                                                 device=self.device)
                recv = self._recv_big[: self.world * slot]
                self._all_gather(recv, self.send[:slot])
                collectives += 1
                if recv.is_cuda:
                    # the collective is ordered on torch's stream, the engine runs on its own:
                    # make the gathered slots visible before admit (phase 1 is fenced by .cpu())
                    torch.cuda.current_stream().synchronize()
            td = time.perf_counter()
            n_cur = self.k.admit(op, labels, self.stamp, rounds, recv, fmt, self.world, slot,
                                 self.rank, self.lo, self.hi, nxt)
            found_total += int(counts.sum())
            if prof is not None:
                prof[2] += td - tc
                prof[3] += time.perf_counter() - td
            cur, nxt = nxt, cur
            rounds += 1
        if labels.is_cuda:
            torch.cuda.synchronize()
        out = {"elapsed_ms": (time.perf_counter() - t0) * 1e3, "supersteps": rounds + 1,
               "pairs_exchanged": found_total, "collectives": collectives, "bitmap_supersteps": dense}
        if prof is not None:
            out["profile_ms"] = {"expand": prof[0] * 1e3, "gather+counts": prof[1] * 1e3,
                                 "big_gather": prof[2] * 1e3, "admit": prof[3] * 1e3}
        return out


class PartitionedRunner:
    """bench.py's N > 1 runner: every rank builds the R-MAT graph, keeps its slice.

    exchange = "rccl"  (default on an nccl group) the C++ superstep loop, collectives issued by the
                       engine on its own RCCL communicator (grx_partitioned_run);
               "hooks" the same C++ loop over host-staged callbacks (gloo rigs sharing one GPU);
               "torch" round 1's Python loop with torch.distributed collectives."""

    def __init__(self, ctx: ea.Context, dist, scale, edge_factor, seed, weight_seed,
                 options: ea.Options | None = None, exchange: str | None = None):
        import torch
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if exchange is None:
            exchange = os.environ.get("GRX_BENCH_EXCHANGE") or \
                ("rccl" if dist.get_backend() == "nccl" else "hooks")
        self.exchange = exchange
        # ONE stream for the engine's kernels, torch's tensor ops and the collectives: the fused
        # superstep needs no cross-stream waits
        self.stream = torch.cuda.Stream(device=ctx.device)
        self.ctx = ctx = ea.Context(ctx.device, stream=self.stream.cuda_stream)
        full = ea.Graph.rmat(ctx, scale, edge_factor, seed, weight_seed, True)
        self._scale, self._edge_factor, self._seed, self._wseed = scale, edge_factor, seed, weight_seed
        self.n, self.nnz = full.n_rows, full.nnz
        self._host = None    # the CPU baseline leg runs at N = 1 only
        ap = full.offsets_to_host()
        self._deg = np.diff(ap).astype(np.int64)
        h = ea._VP()
        lo, hi = C.c_int32(), C.c_int32()
        # the C++ superstep loop runs on slices of the graph's hot-first renumbered copy and hands
        # the labels back in this graph's numbering (grx_graph_partition_hot_first); the
        # torch.distributed loop drives single supersteps and keeps the plain slices
        self.renumbered = exchange in ("rccl", "hooks") and os.environ.get("GRX_HOT_FIRST", "1") != "0"
        if self.renumbered:
            ea._check(ea.load_library().grx_graph_partition_hot_first(
                ctx._h, full._h, self.rank, self.world, C.byref(h), C.byref(lo), C.byref(hi)),
                "grx_graph_partition_hot_first")
        else:
            ea._check(ea.load_library().grx_graph_partition(full._h, self.rank, self.world, C.byref(h),
                                                            C.byref(lo), C.byref(hi)),
                      "grx_graph_partition")
        self.local = ea.Graph(h)
        self.lo, self.hi = lo.value, hi.value
        if not self.renumbered:
            assert [self.lo, self.hi] == partition_bounds(ap, self.world)[self.rank:self.rank + 2]
        full.close()
        dev = f"cuda:{ctx.device}"
        self.depth = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.distance = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.plan = self.trav = None
        self.exchange_note = None
        if exchange in ("rccl", "hooks"):
            # every rank must end up on the same path: agree on whether the attachment worked
            ok, why = 1, ""
            try:
                attach_job(ctx, dist, exchange)
                self.plan = ea.PartitionedPlan(ctx, self.local, self.lo, self.hi, options)
            except Exception as e:   # e.g. ncclCommInitRank refused: keep the torch.distributed loop
                ok, why = 0, repr(e)
            flag = torch.tensor([ok], dtype=torch.int32,
                                device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if self.plan is not None:
                    self.plan.close()
                    self.plan = None
                try:
                    ctx.detach()
                except Exception:
                    pass
                self.exchange_note = f"'{exchange}' attachment failed on some rank ({why}); torch.distributed loop"
                self.exchange = exchange = "torch"
                if self.renumbered:   # that loop drives single supersteps: plain slices, caller's ids
                    self.local.close()
                    full = ea.Graph.rmat(ctx, scale, edge_factor, seed, weight_seed, True)
                    h = ea._VP()
                    ea._check(ea.load_library().grx_graph_partition(full._h, self.rank, self.world, C.byref(h),
                                                                    C.byref(lo), C.byref(hi)),
                              "grx_graph_partition")
                    self.local = ea.Graph(h)
                    self.lo, self.hi = lo.value, hi.value
                    self.renumbered = False
                    full.close()
        if exchange == "torch":
            fused = options is None or options.load_balance == ea.LoadBalance.block_mapped
            with torch.cuda.stream(self.stream):
                self.trav = PartitionedTraversal(HipKernels(ctx, self.local, options), dist,
                                                 self.rank, self.world, self.n, self.lo, self.hi,
                                                 self.local.nnz, dev, fused=fused, stream=self.stream)
        self._torch = torch
        self._deg_dev = torch.from_numpy(self._deg).to(dev)
        self._zero = torch.zeros((), dtype=torch.int64, device=dev)
        self._acc = self._zero.clone()
        self.last = {}

    def verify(self, source: int) -> dict:
        """Outside the timed region: do the ranks' label replicas agree after a partitioned BFS and
        SSSP from `source`, and (rank 0 rebuilds the whole graph) do they equal the single-GPU
        engine's labels bit for bit?  Collective."""
        torch = self._torch
        out = {"source": source}
        sums = []
        for op, labels in ((OP_BFS, self.depth), (OP_SSSP, self.distance)):
            self._traverse(op, source, labels)
            self.stream.synchronize()
            bits = labels.view(torch.int32).to(torch.int64)
            sums.append((int(bits.sum().item()), int((bits * (torch.arange(self.n, device=bits.device) % 1021 + 1)).sum().item())))
        gathered = [None] * self.world
        self.dist.all_gather_object(gathered, sums)
        out["replicas_identical_across_ranks"] = all(g == gathered[0] for g in gathered)
        if self.rank == 0:
            try:
                single = ea.Context(self.ctx.device)
                full = ea.Graph.rmat(single, self._scale, self._edge_factor, self._seed, self._wseed, True)
                d, _ = ea.bfs(single, full, source)
                w, _ = ea.sssp(single, full, source)
                out["bfs_equals_single_gpu"] = bool(torch.equal(d, self.depth))
                out["sssp_bits_equal_single_gpu"] = bool(
                    torch.equal(w.view(torch.int32), self.distance.view(torch.int32)))
                full.close()
                single.close()
            except Exception as e:
                out["single_gpu_check_error"] = repr(e)
        return out

    def close(self) -> None:
        """Release the plan and leave the job (ncclCommDestroy) while the process group still
        exists -- not during interpreter shutdown."""
        if self.plan is not None:
            self.plan.close()
            self.plan = None
        try:
            self.stream.synchronize()
            self.ctx.detach()
        except Exception:
            pass

    def _traverse(self, op: int, source: int, labels) -> dict:
        if self.plan is not None:
            with self._torch.cuda.stream(self.stream):
                return self.plan.run(op, source, labels)
        return self.trav.run(op, source, labels)

    def host_csr(self):
        return self._host

    def global_degrees(self):
        return self._deg

    def _edges(self, labels, unreached):
        """Sum of the out-degrees of the reached vertices, as a device scalar (same stream as the
        traversal: ordered after it and before the next one overwrites the labels)."""
        torch = self._torch
        with torch.cuda.stream(self.stream):
            e = torch.where(labels != unreached, self._deg_dev, self._zero).sum()
            self._acc = self._acc + e
            return e

    def bfs(self, source: int, opts=None) -> int:
        """Returns 0: the traversed-edge count accumulates on the device (no host read in the timed
        region); flush_edges() hands the total over."""
        self.last["bfs"] = self._traverse(OP_BFS, source, self.depth)
        self._edges(self.depth, ea.INT_UNREACHED)
        return 0

    def sssp(self, source: int, opts=None) -> int:
        self.last["sssp"] = self._traverse(OP_SSSP, source, self.distance)
        self._edges(self.distance, ea.FLT_UNREACHED)
        return 0

    def flush_edges(self) -> int:
        self.stream.synchronize()
        total = int(self._acc.item())
        self._acc = self._zero.clone()
        return total

    def bfs_roofline(self, source: int, lb) -> dict:
        st = self._traverse(OP_BFS, source, self.depth)
        edges = int(self._edges(self.depth, ea.INT_UNREACHED).item())
        self.flush_edges()
        reached = int((self.depth != ea.INT_UNREACHED).sum().item())
        nbytes = bfs_algorithmic_bytes(edges, reached)
        achieved = nbytes / (st["elapsed_ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS * self.world,
                "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBPS * self.world), "traffic": None,
                "kernel": "whole partitioned BFS (supersteps incl. all-gather), aggregate over ranks",
                "algorithmic_bytes": nbytes, "elapsed_ms": st["elapsed_ms"],
                "supersteps": st["supersteps"]}

    def detail(self) -> dict:
        return {k: dict(v, rows_owned=self.hi - self.lo, local_edges=self.local.nnz,
                        exchange=self.exchange)
                for k, v in self.last.items()}
