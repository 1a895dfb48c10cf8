// ref_clients_driver.cpp -- the REFERENCE's own algorithm headers, unmodified, on THIS engine.
//
// TEST INFRASTRUCTURE (drop-in evidence, not product).  This translation unit contains no
// reference code: it #includes include/gunrock/algorithms/{bfs,sssp,pr}.hxx where they lie under
// $GRX_REFERENCE_ROOT and wraps gunrock::{bfs,sssp,pr}::run in a C ABI.  With this repository's
// include/ first on the include path, their `#include <gunrock/algorithms/algorithms.hxx>`
// resolves to the MI355X-native engine.  ref_build.sh compiles it for gfx950 into
// oracle/_ref/libgrx_ref_clients.so (git-ignored; travels to the GPU box as a built artefact).
#include GRX_REF_BFS_HXX
#include GRX_REF_SSSP_HXX
#include GRX_REF_PR_HXX

using namespace gunrock;

namespace {
auto make_graph(int n, int nnz, int* ap, int* aj, float* ax) {
  // exactly the call of examples/algorithms/bfs/bfs.cu:46-57 (csr view only)
  return graph::build::from_csr<memory_space_t::device, graph::view_t::csr>(n, n, nnz, ap, aj, ax);
}
}  // namespace

extern "C" int refc_bfs(int n, int nnz, int* d_ap, int* d_aj, float* d_ax, int source,
                        int* d_distances, float* ms) {
  try {
    auto G = make_graph(n, nnz, d_ap, d_aj, d_ax);
    *ms = gunrock::bfs::run(G, source, d_distances, (int*)nullptr);
    return 0;
  } catch (std::exception& e) {
    return -1;
  }
}

extern "C" int refc_sssp(int n, int nnz, int* d_ap, int* d_aj, float* d_ax, int source,
                         float* d_distances, float* ms) {
  try {
    auto G = make_graph(n, nnz, d_ap, d_aj, d_ax);
    *ms = gunrock::sssp::run(G, source, d_distances, (int*)nullptr);
    return 0;
  } catch (std::exception& e) {
    return -1;
  }
}

extern "C" int refc_pr(int n, int nnz, int* d_ap, int* d_aj, float* d_ax, float alpha, float tol,
                       float* d_p, float* ms) {
  try {
    auto G = make_graph(n, nnz, d_ap, d_aj, d_ax);
    *ms = gunrock::pr::run(G, alpha, tol, d_p);
    return 0;
  } catch (std::exception& e) {
    return -1;
  }
}
