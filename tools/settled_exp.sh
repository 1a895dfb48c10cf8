#!/bin/bash
# Runs ON THE GPU BOX: the wide BFS levels of source 0 under the diagnostic builds of
# expand_settled_kernel (GRX_SETTLED_EXP = 1 columns + LDS only, 2 + predicate lookups, 5 + packing
# with an empty functor, 3 + the real functor without the output path), per graph layout.
# Needs essentials_amd/libessentials_amd.{exp1,exp2,exp3,exp5,stats}.so
# (python -m essentials_amd.build --variant expN -D GRX_SETTLED_EXP=N --only capi_bfs).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
for layout in ${LAYOUTS:-generated degree-ordered}; do
  for v in main exp1 exp2 exp5 exp3; do
    lib=$R/essentials_amd/libessentials_amd.so; [ $v != main ] && lib=$R/essentials_amd/libessentials_amd.$v.so
    [ -f $lib ] || continue
    out=$R/gpurun_out/sexp_${layout}_$v
    rm -rf $out
    ESSENTIALS_AMD_LIB=$lib timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/level_probe.py $layout bfs > $out.log 2>&1 || { echo "FAILED $layout $v"; tail -5 $out.log; exit 1; }
    echo "== $layout $v: $(grep 'enact ms' $out.log)"
    python3 tools/trace_levels.py $out | grep -v "publish_counters\|index_kernel\|fill_kernel\|reach_stats"
  done
  if [ -f $R/essentials_amd/libessentials_amd.stats.so ]; then
    echo "== $layout stats"
    ESSENTIALS_AMD_LIB=$R/essentials_amd/libessentials_amd.stats.so timeout -k 10 100 python3 tools/level_probe.py $layout bfs 2>&1 | grep "settled\|enact" | tail -4
  fi
done
