#!/usr/bin/env python3
"""gpurun_out/profiles_raw/ (tools/collect_profiles.sh) -> profiles/rNN_* summaries + latest_pmc.json.

Usage: python3 tools/summarize_profiles.py [round]     (default round 1)"""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
PRE = os.path.join(ROOT, "profiles", f"r{rnd:02d}_bench_")
os.makedirs(os.path.dirname(PRE), exist_ok=True)


def one(pattern):
    hits = sorted(glob.glob(os.path.join(RAW, pattern), recursive=True))
    if not hits:
        sys.exit(f"missing {pattern} under {RAW}")
    return hits[-1]


def short(name):
    """kernel + which client functor it was instantiated for."""
    base = re.sub(r"<.*", "", name).replace("void ", "")
    base = base.split("::")[-1] if "kernels::" in name or "detail::" in name else base[:60]
    client = "-"
    for tag, c in (("bfs_do_enactor_t", "bfs_do"), ("bfs_enactor_t", "bfs"), ("sssp_enactor_t", "sssp"),
                   ("pr_enactor_t", "pr")):
        if tag in name:
            client = c
            break
    return base, client


def bench_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {}


# ---- 1. stats + trace --------------------------------------------------------------------------
shutil.copy(one("stats/**/*kernel_stats.csv"), PRE + "kernel_stats.csv")
rows = list(csv.DictReader(open(one("stats/**/*kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(PRE + "kernel_trace_advance.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "client", "start_ns", "end_ns", "duration_us", "grid_threads", "workgroup",
                "lds_bytes", "vgpr", "sgpr"])
    for r in rows:
        k, c = short(r["Kernel_Name"])
        if c == "-" and "publish_counters" not in k and "degree_sum" not in k:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        w.writerow([k, c, s, e, f"{(e - s) / 1e3:.1f}", r.get("Grid_Size_X", r.get("Grid_Size", "")),
                    r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""),
                    r.get("VGPR_Count", ""), r.get("SGPR_Count", "")])
stats_bench = bench_line(os.path.join(RAW, "stats.json"))

# average advance-kernel time of one push BFS traversal from the trace (what bench's live HIP-event
# measurement must agree with): the command runs steps + warmup traversals plus 3 in the roofline leg
total_us, launches = 0.0, 0
for r in rows:
    k, c = short(r["Kernel_Name"])
    if c == "bfs" and ("block_mapped_kernel" in k or "chunk_kernel" in k):
        total_us += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        launches += 1
n_trav = stats_bench.get("steps", 0) + stats_bench.get("warmup", 0) + 3
per_trav = [total_us / n_trav] * n_trav if n_trav else []


# ---- 2. PMC passes -----------------------------------------------------------------------------
def pmc(passname):
    by_kernel = collections.OrderedDict()
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(one(f"{passname}/**/*counter_collection.csv"))):
        k, c = short(r["Kernel_Name"])
        key = f"{k} [{c}]" if c != "-" else k
        d = by_kernel.setdefault(key, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        dd = disp.setdefault(r["Dispatch_Id"], {"kernel": k, "client": c, "counters": {}})
        dd["counters"][r["Counter_Name"]] = float(r["Counter_Value"])
        d["_dispatches"] = len([1 for x in disp.values() if (x["kernel"], x["client"]) == (k, c)])
    return by_kernel, disp


fetch, _ = pmc("fetch")
write, _ = pmc("write")
fw = {"FETCH_SIZE": {k: {"sum_KB": v["FETCH_SIZE"], "dispatches": int(v["_dispatches"])} for k, v in fetch.items()},
      "WRITE_SIZE": {k: {"sum_KB": v["WRITE_SIZE"], "dispatches": int(v["_dispatches"])} for k, v in write.items()}}
json.dump(fw, open(PRE + "pmc_fetch_write.json", "w"), indent=1)

bfs_bench = bench_line(os.path.join(RAW, "fetch.json"))
# push-BFS traversals in the BFS-only command: steps + warmup + the roofline leg's repeats
launches_per_trav = bfs_bench.get("roofline", {}).get("launches", 0)
bm = fetch.get("block_mapped_kernel [bfs]", {})
trav = int(round(bm.get("_dispatches", 0) / launches_per_trav)) if launches_per_trav else 0
adv_fetch = sum(v["FETCH_SIZE"] for k, v in fetch.items() if k in ("block_mapped_kernel [bfs]", "chunk_kernel [bfs]"))
adv_write = sum(v["WRITE_SIZE"] for k, v in write.items() if k in ("block_mapped_kernel [bfs]", "chunk_kernel [bfs]"))


def known(name, table, counter):
    for k, v in table.items():
        if name in k:
            return v[counter]
    return None


_, busy = pmc("l2busy")
_, hit = pmc("l2hit")
with open(PRE + "pmc_l2_busy.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["dispatch", "kernel", "client", "counter", "value"])
    for table in (busy, hit):
        for did, d in table.items():
            if d["client"] != "bfs":
                continue
            for cn, cv in d["counters"].items():
                w.writerow([did, d["kernel"], d["client"], cn, f"{cv:.0f}"])
bfs_busy = [d["counters"] for d in busy.values() if d["client"] == "bfs"]
tot_busy = sum(c.get("TCC_BUSY_sum", 0) for c in bfs_busy)
tot_cyc = sum(c.get("TCC_CYCLE_sum", 0) for c in bfs_busy)
largest = sorted((d for d in busy.values() if d["client"] == "bfs"),
                 key=lambda d: -d["counters"].get("TCC_CYCLE_sum", 0))[:4]
bfs_hit = [d["counters"] for d in hit.values() if d["client"] == "bfs"]
H = sum(c.get("TCC_HIT_sum", 0) for c in bfs_hit)
M = sum(c.get("TCC_MISS_sum", 0) for c in bfs_hit)
RD = sum(c.get("TCC_READ_sum", 0) for c in bfs_hit)
RS = sum(c.get("TCC_READ_SECTORS_sum", 0) for c in bfs_hit)

roof = bfs_bench.get("roofline", {})
latest = {
    "round": rnd,
    "command": "tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc TCC_BUSY_sum "
               "TCC_CYCLE_sum TCC_REQ_sum TCC_TAG_STALL_sum | --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum "
               "TCC_READ_SECTORS_sum (four separate runs) -- python3 bench.py --steps 2 --warmup 1 "
               "--no-cpu-baseline --no-pagerank --algo bfs",
    "kernels": "block_mapped_kernel + chunk_kernel of the push BFS client, all levels of one traversal",
    "traversals_in_command": trav,
    "FETCH_SIZE_KB_per_traversal": adv_fetch / trav if trav else None,
    "WRITE_SIZE_KB_per_traversal": adv_write / trav if trav else None,
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 128-B requests at 64 B -> x2; "
                  "WRITE_SIZE exact.  Calibrated in the same run on kernels with known bytes (below).  The x2 is "
                  "NOT calibrated for the random 4-B label gathers, so the true read traffic lies between 1x and "
                  "2x FETCH_SIZE.",
    "calibration": {
        "fill_edges_kernel_known_read_KB": None,
        "fill_edges_kernel_FETCH_KB": known("fill_edges_kernel", fetch, "FETCH_SIZE"),
        "emit_kernel_known_write_KB": None,
        "emit_kernel_WRITE_KB": known("emit_kernel", write, "WRITE_SIZE"),
    },
    "traffic_bytes_per_traversal": (2 * adv_fetch + adv_write) * 1024 / trav if trav else None,
    "traffic_bytes_per_traversal_lower_bound": (adv_fetch + adv_write) * 1024 / trav if trav else None,
    "algorithmic_bytes_per_traversal": roof.get("algorithmic_bytes"),
    "trace": {"bfs_advance_us_per_traversal_mean": sum(per_trav) / len(per_trav) if per_trav else None,
              "traversals_seen": len(per_trav),
              "bench_live_kernel_ms": stats_bench.get("roofline", {}).get("kernel_ms")},
    "l2": {
        "busy_frac_all_bfs_advance_dispatches": tot_busy / tot_cyc if tot_cyc else None,
        "hit_rate_bfs_advance": H / (H + M) if H + M else None,
        "read_sectors_per_read": RS / RD if RD else None,
        "largest_dispatches": [
            {"kernel": d["kernel"], **d["counters"],
             "busy_frac": d["counters"].get("TCC_BUSY_sum", 0) / max(d["counters"].get("TCC_CYCLE_sum", 1), 1)}
            for d in largest],
        "reading": "TCC_BUSY/TCC_CYCLE summed over the 128 L2 channels: the fraction of channel-cycles the L2 "
                   "is busy while the kernel runs",
    },
}
# known bytes of the two generator kernels used as calibration (rmat.hip): fill_edges reads 4 B per
# generated edge record field it streams; emit writes 8 B per directed edge
gen = stats_bench.get("config", {})
if gen.get("edges"):
    latest["calibration"]["emit_kernel_known_write_KB"] = gen["edges"] * 8 / 1024
    latest["calibration"]["fill_edges_kernel_known_read_KB"] = gen["edges"] * 4 / 1024
json.dump(latest, open(os.path.join(ROOT, "profiles", "latest_pmc.json"), "w"), indent=1)
print(json.dumps({k: latest[k] for k in ("traversals_in_command", "FETCH_SIZE_KB_per_traversal",
                                         "WRITE_SIZE_KB_per_traversal", "traffic_bytes_per_traversal",
                                         "trace")}, indent=1))
print(json.dumps(latest["l2"], indent=1)[:1500])
