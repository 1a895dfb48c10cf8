#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): the rocprofv3 passes behind
# profiles/ -- one --kernel-trace --stats pass over the default bench, then SEPARATE --pmc passes
# (FETCH_SIZE / WRITE_SIZE / L2 busy / L2 hit) over a BFS-only bench.  Raw output goes to
# gpurun_out/profiles_raw/; tools/summarize_profiles.py turns it into the committed summaries.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_raw
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ALL="python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pagerank"
BENCH_BFS="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pagerank --algo bfs"
echo "[profiles] stats pass"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH_ALL > "$OUT/stats.json" 2> "$OUT/stats.err" || exit 1
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2busy:TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_TAG_STALL_sum" "l2hit:TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_READ_SECTORS_sum"; do
  name=${pass%%:*}; counters=${pass#*:}
  echo "[profiles] pmc pass $name: $counters"
  timeout -k 10 300 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- $BENCH_BFS > "$OUT/$name.json" 2> "$OUT/$name.err" || exit 1
done
echo "[profiles] done"
