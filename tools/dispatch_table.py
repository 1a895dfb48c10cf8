#!/usr/bin/env python3
"""Join the passes of tools/pmc_traffic_dispatch.sh by dispatch id: for every operator kernel of the
LAST traversal in the command, duration (kernel trace), L2 requests per clock and channel, miss
rate, fabric (L2 <-> Infinity Cache / HBM) read and write bytes and their rate.

Read bytes = 32 B x RDREQ_32B + 128 B x RDREQ_128B + 64 B x the rest (calibrated on the gather probe,
DESIGN.md section 5); write bytes = 64 B x WRREQ_64B + 32 B x the rest; atomics reach memory as
TCC_EA0_ATOMIC.  128 L2 channels (16 per XCD x 8), TCC_CYCLE_sum is summed over them.
usage: dispatch_table.py DIR [--json]"""
import collections, csv, glob, json, re, sys

root = sys.argv[1]
WANT = ("expand_settled", "expand_fused", "expand_ranked", "block_mapped_kernel", "chunk_kernel", "classify_hubs",
        "rebuild", "select_range", "gather_probe", "merge_path_kernel", "bucket_kernel", "thread_mapped", "wave_mapped")
CHANNELS = 128


def short(name):
    return re.sub(r"<.*", "", name).split("::")[-1].replace("void ", "").split("(")[0]


def last_traversal(seq):
    """seq: [(dispatch id, kernel name, payload, full name)] in launch order -> the last traversal:
    from the last reset pass of a problem (index_kernel over `...problem_t<...>::reset()`'s lambda)
    on, operator kernels only.  (Round 2 cut at reach_stats_kernel, which the default forms no longer
    launch.)"""
    starts = [i for i, x in enumerate(seq) if "problem_t<" in x[3] and "::reset()" in x[3]]
    lo = starts[-1] if starts else 0
    return [x[:3] for x in seq[lo:] if any(w in x[1] for w in WANT)]


def counters(sub):
    """Counter passes are separate runs of the same deterministic command; their dispatch ids can be
    offset against the trace run's, so passes are aligned by POSITION within the last traversal."""
    fs = sorted(glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True))
    if not fs:
        return []
    vals = collections.OrderedDict()
    names, full = {}, {}
    for r in csv.DictReader(open(fs[-1])):
        d = int(r["Dispatch_Id"])
        vals.setdefault(d, {})
        vals[d][r["Counter_Name"]] = vals[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[d] = short(r["Kernel_Name"])
        full[d] = r["Kernel_Name"]
    return last_traversal([(d, names[d], vals[d], full[d]) for d in sorted(vals)])


trace = sorted(glob.glob(f"{root}/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
seq = last_traversal([(int(r["Dispatch_Id"]), short(r["Kernel_Name"]),
                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"]) for r in rows])
passes = {sub: counters(sub) for sub in ("rdreq", "wrreq", "l2")}
# SSSP's iteration count is timing dependent (which improvement of a vertex lands first), so a
# pass may run one more or one fewer narrow iteration at the END: keep the common prefix
for sub, p in passes.items():
    if not p:
        continue
    k = 0
    while k < min(len(p), len(seq)) and p[k][1] == seq[k][1]:
        k += 1
    if k < len(seq):
        print(f"pass {sub}: kernel sequence agrees with the trace run's for the first {k} of {len(seq)} "
              f"dispatches of the traversal; table cut there", file=sys.stderr)
        seq = seq[:k]
table = []
for i, (d, name, us) in enumerate(seq):
    r = passes["rdreq"][i][2] if passes["rdreq"] else {}
    w = passes["wrreq"][i][2] if passes["wrreq"] else {}
    c = passes["l2"][i][2] if passes["l2"] else {}
    rq, r32, r128 = r.get("TCC_EA0_RDREQ_sum", 0), r.get("TCC_EA0_RDREQ_32B_sum", 0), r.get("TCC_EA0_RDREQ_128B_sum", 0)
    rbytes = 32 * r32 + 128 * r128 + 64 * (rq - r32 - r128)
    wq, w64 = w.get("TCC_EA0_WRREQ_sum", 0), w.get("TCC_EA0_WRREQ_64B_sum", 0)
    wbytes = 64 * w64 + 32 * (wq - w64)
    cyc = c.get("TCC_CYCLE_sum", 0) / CHANNELS
    table.append({"dispatch": d, "kernel": name, "us": us, "l2_req": c.get("TCC_REQ_sum", 0),
                  "l2_miss": c.get("TCC_MISS_sum", 0), "l2_cycles_per_channel": cyc,
                  "req_per_clk_channel": c.get("TCC_REQ_sum", 0) / c["TCC_CYCLE_sum"] if c.get("TCC_CYCLE_sum") else None,
                  "l2_busy": c.get("TCC_BUSY_sum", 0) / c["TCC_CYCLE_sum"] if c.get("TCC_CYCLE_sum") else None,
                  "fabric_read_bytes": rbytes, "fabric_read_requests": rq, "of_128B": r128, "of_32B": r32,
                  "fabric_write_bytes": wbytes, "memory_atomics": w.get("TCC_EA0_ATOMIC_sum", 0),
                  "fabric_TBps": (rbytes + wbytes) / us / 1e6 if us else None})
if "--json" in sys.argv:
    print(json.dumps(table, indent=1))
else:
    print("| dispatch | kernel | us | L2 req (M) | miss | req/clk/channel | L2 busy | fabric read MB (128-B share) | written MB | memory atomics (M) | fabric TB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for t in table:
        miss = t["l2_miss"] / t["l2_req"] if t["l2_req"] else 0
        share = 128 * t["of_128B"] / t["fabric_read_bytes"] if t["fabric_read_bytes"] else 0
        rpc = "%.2f" % t["req_per_clk_channel"] if t["req_per_clk_channel"] is not None else "-"
        busy = "%.2f" % t["l2_busy"] if t["l2_busy"] is not None else "-"
        print(f"| {t['dispatch']} | {t['kernel'][:28]} | {t['us']:.1f} | {t['l2_req'] / 1e6:.2f} | {miss:.0%} | {rpc} | {busy} | "
              f"{t['fabric_read_bytes'] / 1e6:.1f} ({share:.0%}) | {t['fabric_write_bytes'] / 1e6:.1f} | "
              f"{t['memory_atomics'] / 1e6:.3f} | {t['fabric_TBps']:.2f} |")
    tot = sum(t["us"] for t in table)
    print(f"\nsum of listed kernels {tot:.1f} us; fabric read {sum(t['fabric_read_bytes'] for t in table) / 1e9:.3f} GB, "
          f"written {sum(t['fabric_write_bytes'] for t in table) / 1e9:.3f} GB")
