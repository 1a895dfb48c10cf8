#!/usr/bin/env python3
"""bench.py -- traversed edges/sec (MTEPS), BFS + SSSP on synthetic R-MAT, N x MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scale 22] [--lb block_mapped]

One *step* = one BFS followed by one SSSP from the same source (source 0 first, then seeded
random non-isolated vertices) through the HIP path (`libessentials_amd.so`).  The graph is
generated on the GPU and is resident in HBM before the timed region; labels stay on the device.
`value` = (sum over steps of the out-degrees of the vertices each traversal reached, BFS and
SSSP both) / wall time of the K steps (barrier + synchronize on both sides, max over ranks).
Prints ONE JSON line (rank 0).  N > 1: either launched by
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or invoked directly as
`python bench.py --gpus N`: the parent then makes NO GPU call, starts N fresh child processes (one
rank per GPU) with that environment, relays rank 0's line and exits with the children's status.
GRX_BENCH_BACKEND=gloo rehearses the N > 1 path with the ranks sharing one GPU.

Extra objects on the line (prompt section 4):
  roofline     dominant kernel = the BFS advance kernels; achieved = algorithmic bytes of one BFS
               (8 B per traversed edge + 20 B per reached vertex) / summed advance
               kernel time of that BFS measured with HIP events on the engine's stream.
  cpu_baseline the reference's own CPU checker (oracle/_ref, kind "reference") or its C
               restatement (oracle/, kind "port") timed on this box's host, 1 core, one BFS + one
               SSSP from source 0 on the same graph.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # the OpenMP CPU baseline must not spin
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver: dmabuf IPC only (RCCL)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6300.0  # what a copy kernel achieves (same guide): the fabric rate measured bytes run at


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--lb", default="block_mapped")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--weight-seed", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pagerank", action="store_true")
    ap.add_argument("--pagerank-scale", type=int, default=24)
    ap.add_argument("--algo", default="bfs+sssp", choices=["bfs+sssp", "bfs", "sssp"])
    return ap.parse_args()


def pick_sources(deg, steps, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    cand = np.flatnonzero(deg > 0)
    extra = rng.choice(cand, size=max(steps - 1, 0), replace=len(cand) < steps)
    return [0] + [int(x) for x in extra]


def cpu_baseline_strong(Ap, Aj):
    """an honest multi-core baseline beside the reference's heap
    search -- OpenMP level-synchronous top-down BFS on all host cores of this box."""
    from oracle.oracle import Oracle
    import numpy as np
    o = Oracle()
    deg = np.diff(Ap).astype(np.int64)
    best, threads = None, 1
    for _ in range(3):
        d, ms, threads = o.bfs_levelsync(Ap, Aj, 0)
        best = ms if best is None else min(best, ms)
    edges = int(deg[d != 2**31 - 1].sum())
    return {"value": edges / (best * 1e-3) / 1e6, "unit": "MTEPS", "cores": threads, "kind": "port",
            "sample": f"OpenMP level-synchronous BFS from source 0 (not in the reference), best of 3: {best:.0f} ms"}


def cpu_baseline(Ap, Aj, Ax, algo):
    """Reference CPU path on the host cores of this box (1 thread, like the reference)."""
    from oracle.oracle import Oracle, RefOracle, build
    build()
    kind = "reference" if RefOracle.available() else "port"
    import numpy as np
    deg = np.diff(Ap).astype(np.int64)
    edges = 0
    ms = 0.0
    t0 = time.time()
    if RefOracle.available():
        r = RefOracle()
        if "bfs" in algo:
            d, t = r.bfs(Ap, Aj, 0); ms += t; edges += int(deg[d != 2**31 - 1].sum())
        if "sssp" in algo:
            w, t = r.sssp(Ap, Aj, Ax, 0); ms += t; edges += int(deg[w < 3e38].sum())
    else:
        o = Oracle()
        if "bfs" in algo:
            d, t = o.bfs_heap(Ap, Aj, 0); ms += t; edges += int(deg[d != 2**31 - 1].sum())
        if "sssp" in algo:
            w, t = o.sssp_heap(Ap, Aj, Ax, 0); ms += t; edges += int(deg[w < 3e38].sum())
    return {"value": edges / (ms * 1e-3) / 1e6, "unit": "MTEPS", "cores": 1, "kind": kind,
            "sample": f"one {algo} from source 0 on the same R-MAT graph, search loops only "
                      f"({ms:.0f} ms of {time.time() - t0:.0f} s wall)"}


def cpu_baseline_from_n1(a):
    """The CPU baseline is timed at N = 1 only; an N > 1 line carries that figure: from the N = 1
    run of the same session on this box when there was one (it leaves gpurun_out/cpu_baseline_n1.json),
    else from the committed bench line of the round (another box of the same type)."""
    for path, origin in ((os.path.join(ROOT, "gpurun_out", "cpu_baseline_n1.json"), "N=1 run of this session"),
                         (os.path.join(ROOT, "profiles", "latest_bench_line.json"), "committed profiles/latest_bench_line.json")):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        base = rec.get("cpu_baseline")
        if not base:
            continue
        if "scale" in rec:
            same = rec["scale"] == a.scale
        else:
            same = ("scale-%d " % a.scale) in rec.get("config", {}).get("workload", "")
        if same:
            return dict(base, origin=origin, strong=rec.get("cpu_baseline_strong"))
    return None


def pagerank_leg(ea, ctx, a) -> dict:
    """BASELINE configs[3] stand-in, reported beside the headline (outside the timed region): PageRank
    on a DIRECTED R-MAT graph, the push scatter pr.hxx performs and the pull form.  Roofline per
    iteration: 12 B per edge (column, weight, rank word) + 44 B per vertex (offsets,
    previous rank, scale, and the copy / dangling / fill / error passes) against the time of one
    iteration of enact()."""
    try:
        g = ea.Graph.rmat(ctx, a.pagerank_scale, a.edge_factor, a.seed, 0, False)
        nbytes = 12 * g.nnz + 44 * g.n_rows
        out = {"workload": f"PageRank alpha 0.85 tol 1e-6 on directed RMAT scale-{a.pagerank_scale} "
                           f"edgefactor-{a.edge_factor} ({g.nnz} edges)",
               "algorithmic_bytes_per_iteration": nbytes}

        def leg(st):
            ms = st.elapsed_ms / st.iterations
            achieved = nbytes / (ms * 1e-3) / 1e9
            return {"iterations": st.iterations, "ms_per_iteration": ms,
                    "mteps": g.nnz * st.iterations / st.elapsed_ms / 1e3,
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                                 "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None}}
        # pr.hxx's push: the engine sorts a copy of the edge list by destination ONCE per graph
        # (operators/by_destination.hxx) and every iteration and run walks that; the unchanged header
        # walks its first advance on a graph row by row (41 ms) and gets the list from the second on,
        # grx_pagerank's client asks for it before its first
        _, st = ea.pagerank(ctx, g, 0.85, 1e-6)
        first = {"enact_ms": st.elapsed_ms, "iterations": st.iterations,
                 "note": "first run on this graph: the hot-first copy exists already (outside enact); "
                         "enact() holds the sort of its edge list by destination and the iterations"}
        _, st = ea.pagerank(ctx, g, 0.85, 1e-6)
        out["push"] = leg(st)
        out["push"]["first_run"] = first
        out["push"]["note"] = ("pr.hxx's lambda (two lookups by source per edge) through grx_pagerank: on the "
                               "hot-first copy of the graph, ranks handed over in the caller's numbering")
        g.build_in_edges(ctx)
        _, st = ea.pagerank(ctx, g, 0.85, 1e-6, options=ea.Options(direction_optimized=True))
        out["pull"] = leg(st)
        out["pull"]["note"] = ("own client: one pre-multiplied lookup per edge over the same destination-sorted "
                               "list of the hot-first copy")
        # the same on the caller's numbering: what the unchanged pr.hxx gets from the engine
        os.environ["GRX_PR_HOT_FIRST"] = "0"
        try:
            ea.pagerank(ctx, g, 0.85, 1e-6)            # this numbering's sorted list is built here
            _, st = ea.pagerank(ctx, g, 0.85, 1e-6)
            out["push_callers_numbering"] = leg(st)
        finally:
            del os.environ["GRX_PR_HOT_FIRST"]
        g.close()
        return out
    except Exception as e:   # never lose the headline line to the side leg
        return {"error": str(e)}


KERNEL_SOURCES = ("include/gunrock/hip/kernels", "include/gunrock/framework/operators",
                  "include/gunrock/util/math.hxx", "essentials_amd/csrc/clients.hxx")


def kernel_sources_sha() -> str:
    """Hash of the sources the measured kernels are compiled from: what a committed PMC record
    (profiles/latest_pmc.json) must have been collected on to describe the kernels this run times."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        path = os.path.join(ROOT, rel)
        files = [path] if os.path.isfile(path) else sorted(
            os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs)
        for f in files:
            h.update(os.path.relpath(f, ROOT).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def attach_pmc(out: dict, a, world: int) -> None:
    """HBM traffic of the same kernels from the committed rocprofv3 --pmc passes (the counters
    cannot be read from inside the bench): profiles/latest_pmc.json, collected by
    tools/collect_profiles.sh and summarised by tools/summarize_profiles.py, stamped with the git
    head and the kernel-source hash it was collected on.  `traffic_stale` says whether those
    sources have changed since."""
    pmc_path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    if not (world == 1 and a.scale == 22 and a.lb == "block_mapped" and os.path.exists(pmc_path)):
        return
    pmc = json.load(open(pmc_path))
    stale = pmc.get("kernel_sources_sha") != kernel_sources_sha()
    stamp = {"collected_at_git_head": pmc.get("git_head"), "traffic_stale": stale,
             "source": "profiles/latest_pmc.json"}
    for key, roof in (("bfs", out.get("roofline")), ("sssp", out.get("roofline_sssp"))):
        rec = pmc.get("traffic", {}).get(key)
        if not rec or roof is None:
            continue
        roof["traffic"] = rec["bytes_per_traversal"]
        roof["traffic_detail"] = dict(rec, **stamp)
        if roof.get("kernel_ms"):
            gbps = rec["bytes_per_traversal"] / (roof["kernel_ms"] * 1e-3) / 1e9
            roof["measured_frac"] = gbps / HBM_PEAK_GBPS
            roof["measured_frac_of_copy_rate"] = gbps / HBM_COPY_GBPS
            roof["measured_note"] = ("traffic = L2 <-> fabric requests by size (rocprofv3 --pmc TCC_EA0_*): it "
                                     "includes Infinity-Cache hits; measured_frac is against the 8 TB/s peak, "
                                     "measured_frac_of_copy_rate against the ~6.3 TB/s a copy achieves")
    l2 = pmc.get("l2", {})
    if out.get("roofline") is not None:
        out["roofline"]["l2_busy_frac"] = l2.get("busy_frac_bfs_advance")
        out["roofline"]["l2_hit_rate"] = l2.get("hit_rate_bfs_advance")
    pr = out.get("pagerank")
    if isinstance(pr, dict):
        for form in ("push", "pull", "push_callers_numbering"):
            rec = pmc.get("traffic", {}).get("pagerank_" + form)
            if rec and form in pr and "roofline" in pr[form]:
                pr[form]["roofline"]["traffic"] = rec["bytes_per_iteration"]
                pr[form]["roofline"]["traffic_detail"] = dict(rec, **stamp)
                gbps = rec["bytes_per_iteration"] / (pr[form]["ms_per_iteration"] * 1e-3) / 1e9
                pr[form]["roofline"]["measured_frac"] = gbps / HBM_PEAK_GBPS
                pr[form]["roofline"]["measured_frac_of_copy_rate"] = gbps / HBM_COPY_GBPS


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` invoked directly: start N child ranks and relay rank 0's line.
    This process has not touched the GPU (no torch import, no HIP call) and never does: the ranks
    are fresh child processes, not an exec of this one."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(free_port())
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                # one rank failed: the others would wait in a collective for ever -- stop exactly
                # the processes started above, then report
                for q in live:
                    q.kill()
    reader.join(timeout=10)
    # stdout carries the ONE JSON line; anything else a library printed there (gloo's connection
    # notice) goes to stderr
    for line in "".join(chunks).splitlines(keepends=True):
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))
    import numpy as np
    import torch
    import essentials_amd as ea

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    # stdout carries exactly ONE JSON line: whatever libraries print there meanwhile (gloo's
    # connection notice, profiler banners) is sent to stderr until the line is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    # GRX_BENCH_BACKEND=gloo rehearses the N>1 path with several ranks sharing one GPU
    backend = os.environ.get("GRX_BENCH_BACKEND", "nccl")
    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device}"))
        else:
            dist.init_process_group(backend)

    ctx = ea.Context(device)
    lb = ea.LoadBalance[a.lb]
    opts = ea.Options(load_balance=lb)

    if world > 1:
        from essentials_amd.distributed import PartitionedRunner
        runner = PartitionedRunner(ctx, dist, a.scale, a.edge_factor, a.seed, a.weight_seed, opts)
    else:
        from essentials_amd.distributed import SingleRunner
        runner = SingleRunner(ctx, a.scale, a.edge_factor, a.seed, a.weight_seed)

    deg = runner.global_degrees()            # numpy int64 [V] (same on every rank)
    sources = pick_sources(deg, a.steps + a.warmup, a.seed + 99)

    def step(src):
        e = 0
        if "bfs" in a.algo:
            e += runner.bfs(src, opts)
        if "sssp" in a.algo:
            e += runner.sssp(src, opts)
        return e

    def fence():
        torch.cuda.synchronize()
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(sources[i])
    runner.flush_edges()
    if hasattr(runner, "mark"):
        runner.mark()
    fence()
    t0 = time.perf_counter()
    edges = 0
    trace = os.environ.get("GRX_BENCH_TRACE")
    for i in range(a.steps):
        ts = time.perf_counter()
        edges += step(sources[a.warmup + i])
        if trace and rank == 0:
            print(f"[bench] step {i} source {sources[a.warmup + i]}: "
                  f"{(time.perf_counter() - ts) * 1e3:.2f} ms {runner.detail()}", file=sys.stderr)
    fence()
    dt = time.perf_counter() - t0
    edges += runner.flush_edges()   # N > 1: accumulated on the device, no host reads in the timed region
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    timed_detail = runner.detail()
    # ---- roofline of the dominant kernel (BFS advance), outside the timed region ----
    roof = runner.bfs_roofline(sources[a.warmup], lb)
    out = {
        "metric": "traversed edges/sec (MTEPS) BFS+SSSP on RMAT-%d" % a.scale,
        "value": edges / dt / 1e6,
        "unit": "MTEPS",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int32/f32",
        "data": "synthetic",
        "config": {"workload": f"BFS+SSSP on RMAT scale-{a.scale} edgefactor-{a.edge_factor} "
                               f"(symmetrized, {runner.nnz} directed edges), {a.lb} advance, "
                               f"{world}xMI355X",
                   "algo": a.algo, "load_balance": a.lb, "vertices": runner.n, "edges": runner.nnz,
                   "partitioning": "none" if world == 1 else f"1-D vertex ranges x{world}, "
                                   "all-gather of the ranks' output frontiers between supersteps"},
        "roofline": roof,
        "detail": timed_detail,
    }
    if world == 1 and "sssp" in a.algo:
        out["roofline_sssp"] = runner.sssp_roofline(sources[a.warmup], lb)
    if world == 1:
        bfs_mean = out["detail"].get("bfs", {}).get("enact_ms_mean_over_timed_steps") or 0.0
        sssp_mean = out["detail"].get("sssp", {}).get("enact_ms_mean_over_timed_steps") or 0.0
        out["detail"]["outside_enact_ms_per_step"] = out["ms_per_step"] - bfs_mean - sssp_mean
    if world > 1:
        # what ran where: the process group torch sees, the transport the engine's C++ superstep
        # loop used for the data path, one device per rank
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "device": device,
                                      "name": torch.cuda.get_device_name(device),
                                      "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", ""))})
        job = runner.ctx.job_info()
        out["config"].update({"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                              "exchange": runner.exchange, "engine_transport": job["backend"],
                              "engine_world_size": job["world_size"], "devices": devs})
        if runner.exchange_note:
            out["config"]["exchange_note"] = runner.exchange_note
        out["verify"] = runner.verify(sources[a.warmup])
    if world == 1 and a.algo == "bfs+sssp":
        out["reference_clients"] = runner.reference_clients(sources[a.warmup:a.warmup + a.steps], lb)
    if world == 1 and "bfs" in a.algo:
        out["bfs_direction_optimized"] = runner.bfs_direction_optimized(
            sources[a.warmup:a.warmup + min(a.steps, 8)], lb)
    if world == 1 and not a.no_pagerank:
        out["pagerank"] = pagerank_leg(ea, ctx, a)
    if world == 1:
        out["runs_in_process"] = dict(runner.runs)   # what a profile of this command contains
        attach_pmc(out, a, world)
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:   # timed on the host cores at N = 1 only
            Ap, Aj, Ax = runner.host_csr()
            out["cpu_baseline"] = cpu_baseline(Ap, Aj, Ax, a.algo)
            if "bfs" in a.algo:
                out["cpu_baseline_strong"] = cpu_baseline_strong(Ap, Aj)
            try:   # the N > 1 lines of the same session carry it as cpu_baseline_n1
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "cpu_baseline_n1.json"), "w") as f:
                    json.dump({"scale": a.scale, "algo": a.algo, "cpu_baseline": out["cpu_baseline"],
                               "cpu_baseline_strong": out.get("cpu_baseline_strong")}, f)
            except OSError:
                pass
        else:
            out["cpu_baseline"] = None
            if world > 1:   # timed on the host cores at N = 1 only: carry that run's figures
                out["cpu_baseline_n1"] = cpu_baseline_from_n1(a)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        runner.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
