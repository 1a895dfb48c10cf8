#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the wide-level kernels of one BFS (tools/level_profile.py, RMAT-22
# source 0), with and without the settled-destination hint; tools/pmc_dispatch.py prints the last two
# dispatches of the kernel.
# DO NOT add TA_* / TCP_* / TD_* counters here: on this pool every such pass ended in rocprofv3
# aborting (signal 6) only after the command's whole time limit (3 x 200 s lost in round 2).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for f in 1 0; do
  export GRX_SETTLED_FILTER=$f
  K=expand_settled_kernel; [ $f = 0 ] && K=expand_fused_kernel
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD \
      --output-format csv -d $R/gpurun_out/pm_sq$f -- python3 $R/tools/level_profile.py 22 0 > $R/gpurun_out/pm_sq$f.log 2>&1 || exit 1
  echo "settled filter $f: $K"; python3 $R/tools/pmc_dispatch.py $R/gpurun_out/pm_sq$f $K 2
done
