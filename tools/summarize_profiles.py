#!/usr/bin/env python3
"""gpurun_out/profiles_raw/ (tools/collect_profiles.sh) -> profiles/rNN_* summaries + latest_pmc.json.

Usage: python3 tools/summarize_profiles.py [round]     (default round 2; run in the git checkout)

Kernels are attributed to a client by their template arguments (bfs / sssp / pagerank push / pull /
direction-optimised bfs / the gather probe); the bench line of each pass says how many traversals /
iterations of each the command ran (`runs_in_process`, pagerank iterations), which turns sums over
dispatches into per-traversal figures.

HBM read bytes come from the L2's memory-side request counters by size: 32 B x TCC_EA0_RDREQ_32B +
128 B x TCC_EA0_RDREQ_128B + 64 B x the rest -- ONE number, no "x2 or not" bracket.  The guide's
derived FETCH_SIZE (= TCC_EA0_RDREQ x 64 B, i.e. half the bytes of 128-B requests) is kept beside it,
and both are calibrated in the same run on the gather probe (grx_measure_gather_rate: E column
indices streamed once = 4 E bytes known, plus E random 4-B lookups into a 16 MB table that stays
on chip)."""
import collections, csv, glob, hashlib, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
PRE = os.path.join(ROOT, "profiles", f"r{rnd:02d}_bench_")
os.makedirs(os.path.dirname(PRE), exist_ok=True)


def one(pattern):
    # gpurun MERGES new files into gpurun_out/: older runs' files may still lie here -- newest wins
    hits = sorted(glob.glob(os.path.join(RAW, pattern), recursive=True), key=os.path.getmtime)
    if not hits:
        sys.exit(f"missing {pattern} under {RAW}")
    return hits[-1]


ADVANCE = ("block_mapped_kernel", "chunk_kernel", "classify_hubs_kernel", "expand_fused_kernel",
           "expand_settled_kernel", "rebuild_kernel", "wave_chunk_kernel", "pull_probe_kernel",
           "pull_long_kernel", "select_range_kernel")


def attribute(name):
    """(kernel, client) of a dispatch."""
    m = re.search(r"kernels::(\w+)", name) or re.search(r"(\w+)<", name) or re.search(r"(\w+)", name)
    k = m.group(1)
    client = "-"
    for tag, c in (("bfs_do_enactor_t", "bfs_do"), ("bfs_enactor_t", "bfs"), ("sssp_enactor_t", "sssp"),
                   ("pr_pull_enactor_t", "pagerank_pull"), ("pr_enactor_t", "pagerank_push"),
                   ("pr_problem_t", "pagerank_setup")):
        if tag in name:
            client = c
            break
    if k in ("row_group_sum_kernel", "hub_chunk_sum_kernel"):
        client = "pagerank_pull"
    if k == "gather_probe_kernel":
        client = "probe"
    if "by_destination" in name:   # the destination-sorted walk and its one-time build / fingerprint
        k = "by_destination_" + (re.search(r"by_destination::k::(\w+)", name) or m).group(1)
        if client == "-":
            client = "pagerank_push" if k.endswith("expand_kernel") else "pagerank_setup"
    return k, client


def attribute_all(names):
    """attribute() over dispatches in launch order.  A hub pre-pass (no functor in its template
    arguments) belongs to the client of the expansion kernel launched right after it.  A push BFS run
    (its dispatches end with reach_stats_kernel) that has a wide level but no expand_settled_kernel is
    the call_every_edge formulation of bench.py's roofline leg: client "bfs_every_edge".  An SSSP run
    with a bypass_kernel in it is the reference's two-pass formulation (advance + bypass filter,
    grx_options.sssp_two_pass): client "sssp_two_pass" -- round 2 averaged both under "sssp"."""
    out = [attribute(n) for n in names]
    for i, (k, c) in enumerate(out):
        if k == "classify_hubs_kernel" and c == "-":
            for k2, c2 in out[i + 1:i + 3]:
                if k2 in ("expand_fused_kernel", "expand_settled_kernel"):
                    out[i] = (k, c2)
                    break
    # one run = from a problem's reset pass (index_kernel over `...problem_t<...>::reset()`'s lambda)
    # to the next one; round 2 cut at reach_stats_kernel, which the default forms no longer launch
    starts = [i for i, n in enumerate(names)
              if ("problem_t<" in n and "::reset()" in n) or "out_scale_kernel" in n]  # PageRank's reset pass
    for a, b in zip(starts, starts[1:] + [len(out)]):
        run = range(a, b)
        mine = [out[j][0] for j in run if out[j][1] == "bfs"]
        if "expand_fused_kernel" in mine and "expand_settled_kernel" not in mine:
            for j in run:
                if out[j][1] == "bfs":
                    out[j] = (out[j][0], "bfs_every_edge")
        if any(out[j][0] == "bypass_kernel" for j in run):
            for j in run:
                if out[j][1] == "sssp":
                    out[j] = (out[j][0], "sssp_two_pass")
        # the push PageRank run in which the edge list is sorted by destination (one iteration row by
        # row, then the build): not what an iteration costs from then on
        if any(out[j][0] == "by_destination_pack_kernel" for j in run):
            for j in run:
                if out[j][1] == "pagerank_push":
                    out[j] = (out[j][0], "pagerank_push_first_run")
    # bench.py's push runs AFTER its pull leg are on the caller's numbering (GRX_PR_HOT_FIRST=0: what
    # the unchanged pr.hxx gets); the ones before it on the hot-first copy
    seen_pull = False
    for a, b in zip(starts, starts[1:] + [len(out)]):
        clients = {out[j][1] for j in range(a, b)}
        if "pagerank_pull" in clients:
            seen_pull = True
        elif seen_pull:
            for j in range(a, b):
                if out[j][1].startswith("pagerank_push"):
                    out[j] = (out[j][0], out[j][1].replace("pagerank_push", "pagerank_push_callers_numbering"))
    return out


def bench_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {}


# ---- 1. stats + trace of the default bench command -------------------------------------------------
shutil.copy(one("stats/**/*kernel_stats.csv"), PRE + "kernel_stats.csv")
rows = list(csv.DictReader(open(one("stats/**/*kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per_client_us = collections.defaultdict(float)
per_kernel = collections.defaultdict(lambda: [0, 0.0])
with open(PRE + "kernel_trace_advance.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "client", "start_ns", "end_ns", "duration_us", "grid_threads", "workgroup",
                "lds_bytes", "vgpr", "sgpr"])
    for r, (k, c) in zip(rows, attribute_all([r["Kernel_Name"] for r in rows])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        per_kernel[(k, c)][0] += 1
        per_kernel[(k, c)][1] += (e - s) / 1e3
        if c in ("bfs", "bfs_every_edge", "sssp", "sssp_two_pass") and k in ADVANCE:
            per_client_us[c] += (e - s) / 1e3
        if c in ("pagerank_push", "pagerank_pull"):
            per_client_us[c] += (e - s) / 1e3
        if c == "-" and "publish_counters" not in k:
            continue
        w.writerow([k, c, s, e, f"{(e - s) / 1e3:.1f}", r.get("Grid_Size_X", r.get("Grid_Size", "")),
                    r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""),
                    r.get("VGPR_Count", ""), r.get("SGPR_Count", "")])
stats_bench = bench_line(os.path.join(RAW, "stats.json"))
json.dump(stats_bench, open(PRE + "line.json", "w"))
json.dump(stats_bench, open(os.path.join(ROOT, "profiles", "latest_bench_line.json"), "w"))  # bench.py: cpu_baseline_n1
with open(PRE + "kernel_by_client.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "client", "dispatches", "total_us", "mean_us"])
    for (k, c), (n, us) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, f"{us:.1f}", f"{us / n:.2f}"])
runs = stats_bench.get("runs_in_process", {})
pr = stats_bench.get("pagerank", {})
trace = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py",
         "bfs_advance_us_per_traversal": per_client_us["bfs"] / runs["bfs"] if runs.get("bfs") else None,
         "bfs_every_edge_advance_us_per_traversal":
             per_client_us["bfs_every_edge"] / runs["bfs_call_every_edge"] if runs.get("bfs_call_every_edge") else None,
         "sssp_advance_us_per_traversal": per_client_us["sssp"] / runs["sssp"] if runs.get("sssp") else None,
         "sssp_two_pass_advance_us_per_traversal":
             per_client_us["sssp_two_pass"] / runs["sssp_two_pass"] if runs.get("sssp_two_pass") else None,
         "pagerank_push_us_per_iteration":
             per_client_us["pagerank_push"] / pr["push"]["iterations"] if pr.get("push") else None,
         "pagerank_pull_us_per_iteration":
             per_client_us["pagerank_pull"] / pr["pull"]["iterations"] if pr.get("pull") else None,
         "bench_live": {"bfs_kernel_ms": stats_bench.get("roofline", {}).get("kernel_ms"),
                        "bfs_every_edge_kernel_ms": stats_bench.get("roofline", {}).get(
                            "call_every_edge_formulation", {}).get("kernel_ms"),
                        "sssp_kernel_ms": stats_bench.get("roofline_sssp", {}).get("kernel_ms"),
                        "pagerank_push_ms_per_iteration": pr.get("push", {}).get("ms_per_iteration"),
                        "pagerank_pull_ms_per_iteration": pr.get("pull", {}).get("ms_per_iteration")},
         "note": "sssp = the one-pass runs (packed labels); the reference's two-pass formulation is told apart "
                 "by the bypass_kernel of its runs and reported separately"}


# ---- 2. PMC passes ----------------------------------------------------------------------------------
def pmc(passname):
    table = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.OrderedDict()
    raw = list(csv.DictReader(open(one(f"{passname}/**/*counter_collection.csv"))))
    order = sorted({int(r["Dispatch_Id"]): r["Kernel_Name"] for r in raw}.items())
    who = dict(zip((d for d, _ in order), attribute_all([n for _, n in order])))
    for r in raw:
        k, c = who[int(r["Dispatch_Id"])]
        table[(k, c)][r["Counter_Name"]] += float(r["Counter_Value"])
        d = disp.setdefault(r["Dispatch_Id"], {"kernel": k, "client": c, "counters": {}})
        d["counters"][r["Counter_Name"]] = float(r["Counter_Value"])
    return table, disp, bench_line(os.path.join(RAW, passname + ".json"))


rd, _, rd_line = pmc("rdreq")
ft, _, _ = pmc("fetch")
wt, _, _ = pmc("write")
wr, _, _ = pmc("wrreq")
busy, busy_disp, _ = pmc("l2busy")
hit, hit_disp, _ = pmc("l2hit")
pmc_runs = rd_line.get("runs_in_process", {})
pmc_pr = rd_line.get("pagerank", {})


def client_sum(table, client, counter, kernels=None):
    return sum(v.get(counter, 0.0) for (k, c), v in table.items()
               if c == client and (kernels is None or k in kernels))


def read_bytes(client, kernels=None):
    n = client_sum(rd, client, "TCC_EA0_RDREQ_sum", kernels)
    n32 = client_sum(rd, client, "TCC_EA0_RDREQ_32B_sum", kernels)
    n128 = client_sum(rd, client, "TCC_EA0_RDREQ_128B_sum", kernels)
    return 32 * n32 + 128 * n128 + 64 * (n - n32 - n128), {"requests": n, "of_32B": n32, "of_128B": n128}


def traffic(client, per, kernels=None, unit="traversal"):
    if not per:
        return None
    rb, req = read_bytes(client, kernels)
    fetch_kb = client_sum(ft, client, "FETCH_SIZE", kernels)
    write_kb = client_sum(wt, client, "WRITE_SIZE", kernels)
    atom = client_sum(wr, client, "TCC_EA0_ATOMIC_sum", kernels)
    out = {"read_bytes": rb / per, "write_bytes": write_kb * 1024 / per,
           "memory_side_atomics": atom / per,
           "read_requests": {k: v / per for k, v in req.items()},
           "FETCH_SIZE_bytes_raw": fetch_kb * 1024 / per,
           "units_in_command": per}
    out["bytes_per_" + unit] = out["read_bytes"] + out["write_bytes"]
    return out


tr = {"bfs": traffic("bfs", pmc_runs.get("bfs"), ADVANCE),
      "bfs_every_edge": traffic("bfs_every_edge", pmc_runs.get("bfs_call_every_edge"), ADVANCE),
      "sssp": traffic("sssp", pmc_runs.get("sssp"), ADVANCE),
      "sssp_two_pass": traffic("sssp_two_pass", pmc_runs.get("sssp_two_pass"), ADVANCE),
      "pagerank_push": traffic("pagerank_push", pmc_pr.get("push", {}).get("iterations"), None, "iteration"),
      "pagerank_push_callers_numbering": traffic("pagerank_push_callers_numbering",
                                                 pmc_pr.get("push_callers_numbering", {}).get("iterations"), None,
                                                 "iteration"),
      "pagerank_pull": traffic("pagerank_pull", pmc_pr.get("pull", {}).get("iterations"), None, "iteration")}
# calibration on the gather probe: bench.py calls grx_measure_gather_rate twice (agent-scope and plain
# loads), each 1 warm-up + 5 timed passes over the E column indices
probe_disp = int(sum(1 for d in csv.DictReader(open(one("rdreq/**/*counter_collection.csv")))
                     if "gather_probe_kernel" in d["Kernel_Name"] and d["Counter_Name"] == "TCC_EA0_RDREQ_sum"))
E = rd_line.get("config", {}).get("edges")
V = rd_line.get("config", {}).get("vertices")
calib = None
if probe_disp and E:
    rb, req = read_bytes("probe")
    calib = {"dispatches": probe_disp, "known_column_stream_bytes_per_dispatch": 4 * E,
             "label_table_bytes": 4 * V,
             "read_bytes_per_dispatch": rb / probe_disp,
             "FETCH_SIZE_bytes_raw_per_dispatch": client_sum(ft, "probe", "FETCH_SIZE") * 1024 / probe_disp,
             "read_requests_per_dispatch": {k: v / probe_disp for k, v in req.items()},
             "reading": "the size-resolved request counters reproduce the known stream; FETCH_SIZE reports "
                        "about half of it (128-B requests tallied at 64 B), as the guide says"}

# L2 evidence for the BFS advance dispatches
with open(PRE + "pmc_l2.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["pass", "dispatch", "kernel", "client", "counter", "value"])
    for name, table in (("l2busy", busy_disp), ("l2hit", hit_disp)):
        for did, d in table.items():
            if d["client"] in ("bfs", "bfs_every_edge", "sssp", "sssp_two_pass", "pagerank_push", "pagerank_pull", "probe"):
                for cn, cv in d["counters"].items():
                    w.writerow([name, did, d["kernel"], d["client"], cn, f"{cv:.0f}"])


def l2(client):
    b = client_sum(busy, client, "TCC_BUSY_sum")
    cyc = client_sum(busy, client, "TCC_CYCLE_sum")
    h = client_sum(hit, client, "TCC_HIT_sum")
    m = client_sum(hit, client, "TCC_MISS_sum")
    r = client_sum(hit, client, "TCC_READ_sum")
    s = client_sum(hit, client, "TCC_READ_SECTORS_sum")
    return {"busy_frac": b / cyc if cyc else None, "hit_rate": h / (h + m) if h + m else None,
            "read_sectors_per_read": s / r if r else None,
            "requests": client_sum(busy, client, "TCC_REQ_sum"),
            "atomics": client_sum(busy, client, "TCC_ATOMIC_sum")}


from bench import kernel_sources_sha  # noqa: E402  (the same hash bench.py checks at run time)
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "include", "essentials_amd"],
                                capture_output=True, text=True).stdout.strip())
except Exception:
    head, dirty = None, None
latest = {
    "round": rnd,
    "git_head": head, "git_dirty_sources": dirty, "kernel_sources_sha": kernel_sources_sha(),
    "command": "tools/collect_profiles.sh: separate rocprofv3 --pmc passes (TCC_EA0_RDREQ by size | FETCH_SIZE | "
               "WRITE_SIZE | TCC_EA0_WRREQ + atomics | L2 busy | L2 hit) -- python3 bench.py --steps 2 --warmup 1 "
               "--no-cpu-baseline",
    "traffic": tr,
    "calibration_gather_probe": calib,
    "algorithmic": {"bfs_bytes_per_traversal": rd_line.get("roofline", {}).get("algorithmic_bytes"),
                    "sssp_bytes_per_traversal": rd_line.get("roofline_sssp", {}).get("algorithmic_bytes"),
                    "pagerank_bytes_per_iteration": pmc_pr.get("algorithmic_bytes_per_iteration")},
    "trace": trace,
    "l2": {"busy_frac_bfs_advance": l2("bfs")["busy_frac"], "hit_rate_bfs_advance": l2("bfs")["hit_rate"],
           "by_client": {c: l2(c) for c in ("bfs", "bfs_every_edge", "sssp", "sssp_two_pass", "pagerank_push",
                                            "pagerank_pull", "probe")},
           "reading": "TCC_BUSY/TCC_CYCLE summed over the 128 L2 channels and all dispatches of the client"},
}
json.dump(latest, open(os.path.join(ROOT, "profiles", "latest_pmc.json"), "w"), indent=1)
json.dump({"traffic": tr, "calibration_gather_probe": calib}, open(PRE + "pmc_traffic.json", "w"), indent=1)
print(json.dumps({"trace": trace, "traffic": tr, "calibration": calib, "l2": latest["l2"]["by_client"]}, indent=1))
