#!/usr/bin/env bash
# Build the reference's own CPU checkers into oracle/_ref/ (TEST INFRASTRUCTURE).
# Compiles the sources where they lie under the read-only reference tree; nothing
# is copied.  No-op (exit 0) when the reference tree is absent (GPU box).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${GRX_REFERENCE_ROOT:-/root/reference}"
out="$here/_ref"
if [ ! -d "$ref/examples/algorithms/bfs" ]; then
  echo "ref_build: $ref not present; keeping prebuilt $out (if any)"; exit 0
fi
mkdir -p "$out"
hipcc -x hip --cuda-host-only -std=c++17 -O3 -fPIC -shared \
  -DGRX_REF_BFS_CPU="\"$ref/examples/algorithms/bfs/bfs_cpu.hxx\"" \
  -DGRX_REF_SSSP_CPU="\"$ref/examples/algorithms/sssp/sssp_cpu.hxx\"" \
  "$here/ref_driver.cpp" -o "$out/libgrx_ref_oracle.so"
echo "ref_build: built $out/libgrx_ref_oracle.so"

# The reference's unchanged bfs.hxx / sssp.hxx / pr.hxx compiled against THIS repository's
# include/gunrock for gfx950 (GPU tests load it when present).
repo="$(cd "$here/.." && pwd)"
if [ "${GRX_SKIP_REF_CLIENTS:-0}" != "1" ]; then
  hipcc -x hip -std=c++17 -O3 --offload-arch=gfx950 -fPIC -shared \
    -Wno-inconsistent-missing-override -Wno-unused-result \
    -I "$repo/include" \
    -DGRX_REF_BFS_HXX="\"$ref/include/gunrock/algorithms/bfs.hxx\"" \
    -DGRX_REF_SSSP_HXX="\"$ref/include/gunrock/algorithms/sssp.hxx\"" \
    -DGRX_REF_PR_HXX="\"$ref/include/gunrock/algorithms/pr.hxx\"" \
    -DGRX_REF_TC_HXX="\"$ref/include/gunrock/algorithms/tc.hxx\"" \
    "$here/ref_clients_driver.cpp" -o "$out/libgrx_ref_clients.so" -L/opt/rocm/lib -lrccl
  echo "ref_build: built $out/libgrx_ref_clients.so"
  # BASELINE config 3: the unchanged sssp.hxx (which spells block_mapped, sssp.hxx:139) run with
  # the bucketing schedule through the documented compile-time override
  hipcc -x hip -std=c++17 -O3 --offload-arch=gfx950 -fPIC -shared \
    -Wno-inconsistent-missing-override -Wno-unused-result -DGRX_ADVANCE_LB_OVERRIDE=bucketing \
    -I "$repo/include" \
    -DGRX_REF_BFS_HXX="\"$ref/include/gunrock/algorithms/bfs.hxx\"" \
    -DGRX_REF_SSSP_HXX="\"$ref/include/gunrock/algorithms/sssp.hxx\"" \
    -DGRX_REF_PR_HXX="\"$ref/include/gunrock/algorithms/pr.hxx\"" \
    "$here/ref_clients_driver.cpp" -o "$out/libgrx_ref_clients_bucketing.so" -L/opt/rocm/lib -lrccl
  echo "ref_build: built $out/libgrx_ref_clients_bucketing.so"
  # The reference's own example HARNESSES (examples/algorithms/{bfs,sssp,pr}/*.cu, what its CI
  # runs: .github/workflows/ubuntu.yml:52-79), compiled in place and unmodified against this
  # repository's include/.  The reference's include/ is searched AFTER ours and only its
  # algorithms/<algo>.hxx (and, for color, the client-side algorithms/generate/random.hxx it includes)
  # may come from there: the dependency file is checked.
  # bfs / sssp / pr are the hot path's clients; kcore, ppr and bc are the heaviest users of the
  # operators beside it (predicated filters whose predicates have side effects,
  # parallel_for, batch, merge_path over explicit frontiers) and run here purely as drop-in evidence
  for a in bfs sssp pr kcore ppr bc color spmv; do
    hipcc -x hip -std=c++17 -O3 --offload-arch=gfx950 \
      -Wno-inconsistent-missing-override -Wno-unused-result \
      -I "$repo/include" -idirafter "$ref/include" -MD -MF "$out/ref_$a.d" \
      "$ref/examples/algorithms/$a/$a.cu" -o "$out/ref_$a"
    foreign="$(grep -o "$ref/[^ ]*" "$out/ref_$a.d" | sort -u | \
               grep -v -e "^$ref/examples/algorithms/$a/" -e "^$ref/include/gunrock/algorithms/$a.hxx\$" \
                       -e "^$ref/include/gunrock/algorithms/generate/random.hxx\$" || true)"
    if [ -n "$foreign" ]; then
      echo "ref_build: harness $a pulled reference headers other than its algorithm header:"; echo "$foreign"; exit 1
    fi
    rm -f "$out/ref_$a.d"
    echo "ref_build: built $out/ref_$a (harness)"
  done
fi
