// ref_driver.cpp -- C entry points around the REFERENCE's own CPU checkers.
//
// TEST INFRASTRUCTURE.  This translation unit contains no reference code: it
// #includes examples/algorithms/{bfs/bfs_cpu,sssp/sssp_cpu}.hxx where they lie
// under $GRX_REFERENCE_ROOT (default /root/reference) and wraps them in a C ABI.
// ref_build.sh compiles it into oracle/_ref/libgrx_ref_oracle.so (git-ignored,
// travels to the GPU box as a built artefact).  thrust::host_vector, which the
// reference headers use, comes from ROCm's own rocThrust -- not a stand-in.
#include <cstdint>
#include <limits>
#include <thrust/host_vector.h>

#include GRX_REF_BFS_CPU    // <ref>/examples/algorithms/bfs/bfs_cpu.hxx
#include GRX_REF_SSSP_CPU   // <ref>/examples/algorithms/sssp/sssp_cpu.hxx

namespace {
// The shape bfs_cpu::run / sssp_cpu::run expect from their csr_t argument
// (bfs_cpu.hxx:25-27,32 ; sssp_cpu.hxx:27-30,36).
struct host_csr {
  int number_of_rows;
  thrust::host_vector<int> row_offsets;
  thrust::host_vector<int> column_indices;
  thrust::host_vector<float> nonzero_values;
};
host_csr wrap(int n, const int* ap, const int* aj, const float* ax) {
  host_csr c;
  c.number_of_rows = n;
  c.row_offsets.assign(ap, ap + n + 1);
  c.column_indices.assign(aj, aj + ap[n]);
  if (ax) c.nonzero_values.assign(ax, ax + ap[n]);
  return c;
}
}  // namespace

extern "C" float ref_bfs_cpu(int n, const int* ap, const int* aj, int source, int* depth) {
  host_csr c = wrap(n, ap, aj, nullptr);
  return bfs_cpu::run<host_csr, int, int>(c, source, depth, (int*)nullptr);
}

extern "C" float ref_sssp_cpu(int n, const int* ap, const int* aj, const float* ax, int source,
                              float* dist) {
  host_csr c = wrap(n, ap, aj, ax);
  return sssp_cpu::run<host_csr, int, int, float>(c, source, dist, (int*)nullptr);
}
