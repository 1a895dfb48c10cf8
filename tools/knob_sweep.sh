#!/bin/bash
# Runs ON THE GPU BOX: mean BFS / SSSP times of six sources (tools/source_mean.py) under one environment
# knob at a time; the first and the last line are the defaults (their difference is the box's noise).
for cfg in "GRX_DUMMY=0" "GRX_SETTLED_MIN_WORK=262144" "GRX_SETTLED_MIN_WORK=4194304" "GRX_FUSED_MIN_SLOTS=8192" "GRX_FUSED_MIN_SLOTS=131072" "GRX_LABEL_SCAN_MIN_WORK=4194304" "GRX_LABEL_SCAN_MIN_WORK=33554432" "GRX_HUB_THRESHOLD=128" "GRX_HUB_THRESHOLD=512" "GRX_DUMMY=1"; do
  echo -n "$cfg: "; env $cfg python3 tools/source_mean.py 2>&1 | tail -1
done
