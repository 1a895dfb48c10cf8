"""The multi-GPU superstep protocol on CPU ranks: world_size 2 (and 3) over gloo.

essentials_amd.distributed.PartitionedTraversal is the production host loop (exchange format,
small-slot / big-slot all-gather, termination, frontier double buffering).  Here its `kernels`
object is a numpy restatement of grx_partitioned_expand / grx_partitioned_admit (TEST ONLY --
the product's kernels are HIP and have no CPU form), so that the protocol itself runs in real
separate processes over a real process group.  The result on every rank must equal the oracle's
single-process answer.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from essentials_amd.distributed import (OP_BFS, OP_SSSP, RECV_LEVEL_BITMAP, RECV_REPLICA_MIN,  # noqa: E402
                                        PartitionedPageRank, PartitionedTraversal, partition_bounds)


class NumpyKernels:
    """CPU stand-in for HipKernels with the SAME contracts (include/essentials_amd.h)."""

    def __init__(self, Ap, Aj, Ax, lo, hi):
        # local CSR: all V rows, rows outside [lo,hi) empty (what grx_graph_partition builds)
        base, top = Ap[lo], Ap[hi]
        self.ap = (np.clip(Ap, base, top) - base).astype(np.int64)
        self.aj = Aj[base:top]
        self.ax = Ax[base:top]

    def expand(self, op, labels, iparam, frontier, n_frontier, scratch, sent, send):
        lab = labels.numpy()
        snt = sent.numpy()
        found = []
        for v in frontier.numpy()[:n_frontier]:
            for e in range(self.ap[v], self.ap[v + 1]):
                d = self.aj[e]
                new = (iparam + 1) if op == OP_BFS else np.float32(lab[v] + self.ax[e])
                if new < lab[d]:
                    lab[d] = new
                    if op == OP_BFS or snt[d] != iparam:   # SSSP: packed once per vertex and superstep
                        snt[d] = iparam
                        found.append(d)
        s = send.numpy()
        s[0] = len(found)
        k = min(len(found), len(s) - 1)
        if k:
            f = np.array(found[:k], np.int64)
            bits = lab[f].view(np.uint32).astype(np.int64)
            s[1:1 + k] = (bits << 32) | f
        return len(found)

    def step(self, op, labels, stamp, sent, rnd, recv, fmt, world, slot, rank, lo, hi, frontier,
             fcount, scratch, send, snapshot=None):
        """grx_partitioned_step: admit the gather of superstep rnd - 1, snapshot, advance, pack."""
        if recv is not None:
            fcount[0] = self.admit(op, labels, stamp, rnd - 1, recv, fmt, world, slot, rank, lo, hi,
                                   frontier)
        if snapshot is not None:
            snapshot.numpy()[lo:hi] = labels.numpy()[lo:hi]
        self.expand(op, labels, rnd, frontier, int(fcount[0]), scratch, sent, send)

    def pr_scatter(self, alpha, p, scale, compute_scale, partial, lo, hi):
        """grx_pagerank_partitioned_scatter on the local rows."""
        n = len(self.ap) - 1
        sc, pr, out = scale.numpy(), p.numpy(), partial.numpy()
        src = np.repeat(np.arange(n), np.diff(self.ap))
        if compute_scale:
            tot = np.zeros(n, np.float32)
            np.add.at(tot, src, self.ax)
            sc[:] = np.where(tot != 0, np.float32(alpha) / np.where(tot != 0, tot, 1), 0).astype(np.float32)
        out[:n] = np.bincount(self.aj, weights=(pr[src] * sc[src] * self.ax).astype(np.float64),
                              minlength=n).astype(np.float32)
        owned = np.arange(lo, hi)
        out[n] = np.float32(alpha) * pr[owned][sc[owned] == 0].sum(dtype=np.float64)

    def level_bitmap(self, depth, level, words):
        """grx_partitioned_level_bitmap: bit v = (depth[v] == level)."""
        d = depth.numpy()
        bits = np.zeros(len(words) * 64, np.uint8)
        bits[:len(d)] = d == level
        words.numpy()[:] = np.packbits(bits, bitorder="little").view(np.int64)

    def admit(self, op, labels, stamp, rnd, recv, fmt, world, slot, rank, lo, hi, nxt):
        lab, st, out = labels.numpy(), stamp.numpy(), nxt.numpy()
        if fmt == RECV_REPLICA_MIN:       # recv = the snapshot; labels were all-reduced (MIN)
            mine = np.nonzero(lab[lo:hi] < recv.numpy()[lo:hi])[0] + lo
            out[:len(mine)] = mine
            return len(mine)
        if fmt == RECV_LEVEL_BITMAP:
            assert op == OP_BFS
            r = recv.numpy().reshape(world, slot)
            union = np.bitwise_or.reduce(r, axis=0)
            bits = np.unpackbits(union.view(np.uint8), bitorder="little")[:len(lab)].astype(bool)
            lab[bits & (lab > rnd + 1)] = rnd + 1
            mine = np.nonzero(bits)[0]
            mine = mine[(mine >= lo) & (mine < hi)]
            out[:len(mine)] = mine
            return len(mine)
        r = recv.numpy().reshape(world, slot)
        n = 0
        for p in range(world):
            cnt = int(r[p, 0])
            for w in r[p, 1:1 + cnt]:
                v = int(w & 0xFFFFFFFF)
                l = np.array([(int(w) >> 32) & 0xFFFFFFFF], np.uint32).view(lab.dtype)[0]
                fresh = True if p == rank else bool(l < lab[v])
                if p != rank and fresh:
                    lab[v] = l
                if fresh and lo <= v < hi and (op == OP_BFS or st[v] != rnd):
                    st[v] = rnd
                    out[n] = v; n += 1
        return n


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scale, seed, wseed, sources, small_slot, out_dir, fused=True,
            dense_threshold=None, replica_threshold=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    o = Oracle()
    n, Ap, Aj, Ax = o.rmat_csr(scale, 8, seed, wseed)
    b = partition_bounds(Ap, world)
    lo, hi = b[rank], b[rank + 1]
    k = NumpyKernels(Ap, np.ascontiguousarray(Aj), np.ascontiguousarray(Ax), lo, hi)
    trav = PartitionedTraversal(k, dist, rank, world, n, lo, hi, int(Ap[hi] - Ap[lo]), "cpu",
                                small_slot=small_slot, fused=fused, dense_threshold=dense_threshold,
                                replica_threshold=replica_threshold)
    ok = True
    why = []
    for s in sources:
        depth = torch.empty(n, dtype=torch.int32)
        st = trav.run(OP_BFS, s, depth)
        want, _ = o.bfs_heap(Ap, Aj, s)
        if not (depth.numpy() == want).all():
            ok = False; why.append(f"bfs {s}: {int((depth.numpy() != want).sum())} wrong")
        distance = torch.empty(n, dtype=torch.float32)
        st2 = trav.run(OP_SSSP, s, distance)
        if (replica_threshold is not None and fused and int((want != 2**31 - 1).sum()) > 8 * replica_threshold
                and not st2.get("allreduce_supersteps")):
            ok = False; why.append(f"no all-reduce superstep for {s}: {st2}")
        wantw, _ = o.sssp_heap(Ap, Aj, Ax, s)
        if not (distance.numpy().view(np.uint32) == wantw.view(np.uint32)).all():
            ok = False; why.append(f"sssp {s} wrong")
        reached = int((want != 2**31 - 1).sum())
        if st["collectives"] < st["supersteps"] or (reached > 1 and st["supersteps"] < 2):
            ok = False; why.append(f"stats {s}: {st}")
        if dense_threshold is not None and reached > 8 * dense_threshold and not st["bitmap_supersteps"]:
            ok = False; why.append(f"no bitmap superstep for {s}: {st}")
    # PageRank on the same partition against the oracle's restatement of pr.hxx
    p = torch.empty(n, dtype=torch.float32)
    st = PartitionedPageRank(k, dist, rank, world, n, lo, hi, "cpu").run(p, 0.85, 1e-6)
    want, it = o.pagerank(Ap, Aj, Ax, 0.85, 1e-6)
    if not np.allclose(p.numpy(), want, rtol=2e-4, atol=1e-9) or abs(st["iterations"] - it) > 1:
        ok = False; why.append(f"pagerank: max rel err {np.max(np.abs(p.numpy() - want) / want):.2e}, "
                               f"{st['iterations']} vs {it} iterations")
    open(os.path.join(out_dir, f"rank{rank}.ok" if ok else f"rank{rank}.bad"), "w").write(str(why))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,small_slot,fused,dense", [(2, None, True, None), (2, 8, True, None),
                                                          (3, 64, True, None), (2, 8, False, None),
                                                          (2, 8, True, 4), (3, 8, False, 4)])
def test_partitioned_traversal_gloo(tmp_path, world, small_slot, fused, dense):
    _spawn(tmp_path, world, small_slot, fused, dense, None)


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_sssp_allreduce_supersteps_gloo(tmp_path, world):
    """Supersteps with more than 4 finds on some rank combine the distance replicas with an
    all-reduce (MIN) and admit against the pre-advance snapshot (GRX_RECV_REPLICA_MIN)."""
    _spawn(tmp_path, world, 8, True, 4, 4)


def _spawn(tmp_path, world, small_slot, fused, dense, replica):
    """small_slot 8 / 64 forces the second (big-slot) all-gather on the wide levels; fused=True is
    the one-call-per-superstep loop bench.py uses, False the two-call (expand / admit) loop;
    dense=4 makes BFS supersteps with more than 4 finds exchange level bitmaps."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 9, 4, 7, [0, 5, 300], small_slot, str(tmp_path), fused, dense,
                            replica),
             nprocs=world, join=True)
    names = sorted(os.listdir(tmp_path))
    notes = {f: open(os.path.join(tmp_path, f)).read() for f in names}
    assert names == [f"rank{r}.ok" for r in range(world)], notes


def test_partition_bounds_balance_edges():
    from oracle.oracle import Oracle
    o = Oracle()
    n, Ap, Aj, Ax = o.rmat_csr(12, 16, 1, 0)
    for world in (1, 2, 4, 8):
        b = partition_bounds(Ap, world)
        assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))
        per = [int(Ap[b[k + 1]] - Ap[b[k]]) for k in range(world)]
        assert sum(per) == int(Ap[n])
        # no rank exceeds its fair share by more than the largest single row
        assert max(per) <= Ap[n] / world + np.diff(Ap).max()
