#!/bin/bash
# Runs ON THE GPU BOX: PER-DISPATCH fabric traffic of the wide BFS levels and SSSP iterations
# (VERDICT r2 item 1).  One --kernel-trace pass for durations, then SEPARATE --pmc passes (never
# combined with a trace) over the same deterministic command; tools/dispatch_table.py joins them by
# dispatch id.   usage: pmc_traffic_dispatch.sh TAG "python3 tools/level_profile.py 22 0"
set -o pipefail
TAG=$1; shift
CMD="$*"
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/dispatch_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || exit 1
for pass in "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" \
            "wrreq:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" \
            "l2:TCC_REQ_sum TCC_MISS_sum TCC_CYCLE_sum TCC_BUSY_sum"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout -k 10 200 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- $CMD > "$OUT/$name.log" 2>&1 || exit 1
done
python3 $R/tools/dispatch_table.py "$OUT" > "$OUT/table.md" && cat "$OUT/table.md"
