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
#ifdef GRX_REF_TC_HXX
#include GRX_REF_TC_HXX  // triangle counting: block_mapped graph -> none advance + set intersections;
                         // the only algorithm whose results the reference's unit tests pin
                         // (unittests/algorithms/tc.cuh:19-93)
#endif

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

// ---- the same unchanged headers as ONE RANK of a vertex-partitioned job -------------------------
// G is the rank's slice (global ids, rows outside [lo, hi) empty: grx_graph_partition); the
// context is attached to the job, so enactor_t::enact() exchanges the frontiers between supersteps
// (include/gunrock/framework/partitioned.hxx).  transport 0: host callbacks (all_gather /
// all_reduce as in include/essentials_amd.h); 1: RCCL with the 128-byte id in `id128`.
typedef int (*refc_all_gather_fn)(void*, const void*, void*, unsigned long long, void*);
typedef int (*refc_all_reduce_fn)(void*, void*, unsigned long long, int, int, void*);

namespace {
struct hooks_t {
  refc_all_gather_fn ag;
  refc_all_reduce_fn ar;
};
std::shared_ptr<gcuda::multi_context_t> job_context(int rank, int world, int lo, int hi, int transport,
                                                    const void* id128, refc_all_gather_fn ag,
                                                    refc_all_reduce_fn ar) {
  auto mc = std::make_shared<gcuda::multi_context_t>(0);
  if (transport == 1) {
    mc->attach_job(rank, world, gcuda::rccl::make(rank, world, id128, 0));
  } else {
    gcuda::collective_table_t t;
    t.state = new hooks_t{ag, ar};
    t.name = "hooks";
    t.stream_ordered = false;
    t.all_gather = [](void* s, const void* a, void* b, std::size_t n, hipStream_t st) -> int {
      return static_cast<hooks_t*>(s)->ag(nullptr, a, b, n, (void*)st);
    };
    t.all_reduce = [](void* s, void* b, std::size_t n, int d, int o, hipStream_t st) -> int {
      return static_cast<hooks_t*>(s)->ar(nullptr, b, n, d, o, (void*)st);
    };
    t.destroy = [](void* s) { delete static_cast<hooks_t*>(s); };
    mc->attach_job(rank, world, t);
  }
  mc->set_owned_rows(lo, hi);
  return mc;
}
}  // namespace

extern "C" int refc_bfs_job(int n, int nnz, int* d_ap, int* d_aj, float* d_ax, int source,
                            int* d_distances, float* ms, int rank, int world, int lo, int hi,
                            int transport, const void* id128, refc_all_gather_fn ag,
                            refc_all_reduce_fn ar) {
  try {
    auto G = make_graph(n, nnz, d_ap, d_aj, d_ax);
    *ms = gunrock::bfs::run(G, source, d_distances, (int*)nullptr,
                            job_context(rank, world, lo, hi, transport, id128, ag, ar));
    return 0;
  } catch (std::exception& e) {
    std::fprintf(stderr, "refc_bfs_job: %s\n", e.what());
    return -1;
  }
}

extern "C" int refc_sssp_job(int n, int nnz, int* d_ap, int* d_aj, float* d_ax, int source,
                             float* d_distances, float* ms, int rank, int world, int lo, int hi,
                             int transport, const void* id128, refc_all_gather_fn ag,
                             refc_all_reduce_fn ar) {
  try {
    auto G = make_graph(n, nnz, d_ap, d_aj, d_ax);
    *ms = gunrock::sssp::run(G, source, d_distances, (int*)nullptr,
                             job_context(rank, world, lo, hi, transport, id128, ag, ar));
    return 0;
  } catch (std::exception& e) {
    std::fprintf(stderr, "refc_sssp_job: %s\n", e.what());
    return -1;
  }
}

/// pr.hxx declares no replica combiner: a partitioned run must refuse ("replicas only").
extern "C" int refc_pr_job_refused(int n, int nnz, int* d_ap, int* d_aj, float* d_ax, float* d_p,
                                   const void* id128) {
  try {
    auto G = make_graph(n, nnz, d_ap, d_aj, d_ax);
    gunrock::pr::run(G, 0.85f, 1e-6f, d_p, job_context(0, 1, 0, n, 1, id128, nullptr, nullptr));
    return 0;
  } catch (std::exception& e) {
    return std::string(e.what()).find("replica") != std::string::npos ? 1 : -1;
  }
}

#ifdef GRX_REF_TC_HXX
extern "C" int refc_tc(int n, int nnz, int* d_ap, int* d_aj, float* d_ax, int* d_vertex_triangles,
                       unsigned long long* total, float* ms) {
  try {
    auto G = make_graph(n, nnz, d_ap, d_aj, d_ax);
    std::size_t t = 0;
    *ms = gunrock::tc::run(G, true, d_vertex_triangles, &t);
    *total = (unsigned long long)t;
    return 0;
  } catch (std::exception& e) {
    std::fprintf(stderr, "refc_tc: %s\n", e.what());
    return -1;
  }
}
#endif
