#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): the rocprofv3 passes behind
# profiles/.
#   stats   --kernel-trace --stats over the DEFAULT bench command (python3 bench.py): the kernel
#           durations bench.py's live HIP-event numbers must agree with (BFS, SSSP, PageRank).
#   PMC     SEPARATE --pmc passes (never combined with a trace) over a shorter bench that still
#           runs every leg: HBM reads by request size (TCC_EA0_RDREQ 32 / 64 / 128 B), the guide's
#           FETCH_SIZE / WRITE_SIZE, writes and memory-side atomics, L2 busy / hit.
# Raw output goes to gpurun_out/profiles_raw/; tools/summarize_profiles.py (run where git is) turns
# it into the committed summaries under profiles/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_raw
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_DEFAULT="python3 $ROOT/bench.py"
BENCH_PMC="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
echo "[profiles] stats pass: $BENCH_DEFAULT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH_DEFAULT > "$OUT/stats.json" 2> "$OUT/stats.err" || exit 1
for pass in "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" \
            "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
            "wrreq:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" \
            "l2busy:TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_ATOMIC_sum" \
            "l2hit:TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_READ_SECTORS_sum"; do
  name=${pass%%:*}; counters=${pass#*:}
  echo "[profiles] pmc pass $name: $counters"
  timeout -k 10 400 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- $BENCH_PMC > "$OUT/$name.json" 2> "$OUT/$name.err" || exit 1
done
echo "[profiles] done"
