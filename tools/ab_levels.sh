#!/bin/bash
# Runs ON THE GPU BOX: A/B of one environment knob on the same box -- per-kernel times of the last
# BFS and SSSP traversal of source 0 (tools/level_probe.py under rocprofv3 --kernel-trace) and the
# mean of six sources (tools/source_mean.py).   usage: ab_levels.sh NAME=VALUE_A NAME=VALUE_B ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
for setting in "$@"; do
  tag=$(echo "$setting" | tr '= ' '__')
  for algo in bfs sssp; do
    out=$R/gpurun_out/ab_${tag}_$algo; rm -rf $out
    env_name=${setting%%=*}; env_val=${setting#*=}
    export $env_name="$env_val"
    timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/level_probe.py generated $algo > $out.log 2>&1 || { echo "FAILED $setting $algo"; tail -5 $out.log; exit 1; }
    echo "== $setting $algo: $(grep 'enact ms' $out.log)"
    python3 tools/trace_levels.py $out | grep -v "publish_counters\|fill_kernel\|reach_stats\|index_kernel"
  done
  python3 tools/source_mean.py
  unset $env_name
done
