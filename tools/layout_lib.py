"""Relabelled copies of an engine graph for layout experiments (tools/reorder_probe.py,
tools/level_probe.py): same graph, vertex ids permuted, rows keep their edge order."""
import numpy as np, torch
import essentials_amd as ea


def relabelled(ctx, g, layout, seed=5):
    """-> (graph, rank) with rank[old id] = new id (None for the generated layout)."""
    if layout == "generated":
        return g, None
    dev = "cuda"
    Ap, Aj, Ax = g.to_host()
    ap = torch.from_numpy(Ap.astype(np.int64)).to(dev)
    aj = torch.from_numpy(Aj).to(dev)
    ax = torch.from_numpy(Ax).to(dev)
    deg = ap[1:] - ap[:-1]
    n = g.n_rows
    if layout == "degree-ordered":
        order = torch.argsort(deg, descending=True, stable=True)
    elif layout == "scrambled":
        order = torch.from_numpy(np.random.default_rng(seed).permutation(n)).to(dev)
    else:
        raise ValueError(layout)
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[order] = torch.arange(n, device=dev)
    ndeg = deg[order]
    nap = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    nap[1:] = torch.cumsum(ndeg, 0)
    row_of = torch.repeat_interleave(torch.arange(n, device=dev), ndeg)
    src_e = ap[order][row_of] + (torch.arange(aj.numel(), device=dev) - nap[row_of])
    del row_of
    naj = rank[aj[src_e].long()].int().contiguous()
    nax = ax[src_e].contiguous()
    del src_e
    torch.cuda.synchronize()
    return ea.Graph.from_device_csr(nap.int().contiguous(), naj, nax), rank
