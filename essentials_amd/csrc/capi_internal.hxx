/**
 * @file capi_internal.hxx
 * @brief Shared definitions of the C-ABI translation units (not installed).
 */
#pragma once

#include <algorithm>
#include <cfloat>
#include <climits>
#include <memory>
#include <mutex>
#include <vector>
#include <string>

#include <gunrock/algorithms/algorithms.hxx>
#include <gunrock/hip/algorithms.hxx>

#include "../../include/essentials_amd.h"

namespace essentials_amd {

using namespace gunrock;

using vertex_t = int32_t;
using edge_t = int32_t;
using weight_t = float;

using graph_type = decltype(graph::build::from_csr<memory_space_t::device, graph::view_t::csr>(
    vertex_t(0), vertex_t(0), edge_t(0), (edge_t*)nullptr, (vertex_t*)nullptr, (weight_t*)nullptr));
using frontier_type = frontier::frontier_t<vertex_t, edge_t>;

std::string& last_error();

/// Run `body`, translating the C++ surface's exceptions into status codes.
template <typename F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const error::exception_t& e) {
    last_error() = e.what();
    return GRX_ERR_RUNTIME;
  } catch (const std::exception& e) {
    last_error() = e.what();
    return GRX_ERR_RUNTIME;
  } catch (...) {
    last_error() = "unknown exception";
    return GRX_ERR_RUNTIME;
  }
}

inline int invalid(const char* why) {
  last_error() = why;
  return GRX_ERR_INVALID_ARGUMENT;
}
inline int unsupported(const char* why) {
  last_error() = why;
  return GRX_ERR_UNSUPPORTED;
}

}  // namespace essentials_amd

struct grx_context_s {
  std::shared_ptr<gunrock::gcuda::multi_context_t> mc;
  int device = 0;
  unsigned long long pending_sequence = 0;  // counters hand-off of an enqueue-only call
  // partitioned supersteps: what the job found in the PREVIOUS superstep (all ranks; -1 unknown) --
  // the size class of the frontier the next grx_partitioned_step expands -- and the settled bitmap
  // its wide BFS supersteps rebuild (operators/settled.hxx)
  long long superstep_finds_hint = -1;
  long long superstep_found_so_far = 0;  // all ranks, all earlier supersteps of the run
  gunrock::operators::advance::settled_filter_t<int32_t> superstep_settled;
  gunrock::hip::device_array_t<unsigned short> superstep_bound16;  // SSSP: 2-byte distance bounds
  gunrock::gcuda::standard_context_t& single() { return *mc->get_context(0); }
};

struct grx_graph_s;
namespace essentials_amd {
/// max over rows of (offsets[i + 1] - offsets[i]) of a device CSR; synchronous (capi_core.hip).
unsigned long long reduce_max_degree(const int32_t* d_row_offsets, int32_t n_rows);
/// A pull traversal is about to walk `g`'s in-edges: a graph without an attached transpose must
/// be its own transpose.  Verifies that once on the device when nobody has said (builds the
/// transpose, compares it with the column-sorted CSR); GRX_ERR_UNSUPPORTED for a directed graph
/// (capi_core.hip).  Call inside guarded().
int ensure_can_pull(grx_context_s* ctx, grx_graph_s* g);
/// The hot-first renumbered copy of `g` (graph::build::hot_first) the traversals run on, built on
/// first use; nullptr when the handle or GRX_HOT_FIRST says no, or the graph is too small for it
/// to matter (capi_core.hip).  Call inside guarded().
/// `csr_only`: the caller walks out-edges only (PageRank), so a graph with an attached transpose --
/// whose copy would have none -- may run on its copy too.
grx_graph_s* hot_copy(grx_context_s* ctx, grx_graph_s* g, bool csr_only = false);
}  // namespace essentials_amd

struct grx_graph_s {
  grx_graph_s() {
    // a graph's arrays live long and have odd sizes: when the handle dies they go straight back to
    // the device instead of being parked for reuse (hip::block_cache_t)
    ap.set_parking(false);
    aj.set_parking(false);
    ax.set_parking(false);
  }
  int32_t n_rows = 0, n_cols = 0;
  int64_t nnz = 0;
  // owning storage (empty for a view over caller memory)
  gunrock::hip::device_array_t<int32_t> ap, aj;
  gunrock::hip::device_array_t<float> ax;
  const int32_t* d_ap = nullptr;
  const int32_t* d_aj = nullptr;
  const float* d_ax = nullptr;
  // optional in-edge arrays (grx_graph_build_in_edges): marks the graph directed
  std::unique_ptr<gunrock::graph::transposed_t<int32_t, int32_t, float>> in_edges;

  // largest out-degree, reduced on first use (the CSR arrays of a handle do not change)
  mutable unsigned long long max_degree = 0;
  mutable bool max_degree_known = false;
  // is the CSR its own transpose (an undirected graph)?  Known to the builders that symmetrise
  // (R-MAT, symmetric Matrix Market files); otherwise verified on the device the first time a
  // PULL traversal asks (essentials_amd::ensure_can_pull)
  enum symmetry_t { symmetry_unknown = 0, symmetric = 1, asymmetric = 2 };
  mutable int symmetry = symmetry_unknown;
  // Hot-first renumbered copy (include/gunrock/graph/reorder.hxx) that grx_bfs / grx_sssp run on;
  // labels are delivered in the caller's numbering.  -1: automatic (GRX_HOT_FIRST, size), 0: off,
  // 1: on (grx_graph_hot_first).
  int hot_first = -1;
  std::mutex hot_mutex;  // the copy is built on first use: handles may be shared between host threads
  std::unique_ptr<grx_graph_s> hot;
  gunrock::hip::device_array_t<int32_t> hot_vertex_of;  // device: caller's id of a renumbered vertex
  std::vector<int32_t> hot_rank_of;                     // host: renumbered id of a caller's vertex
  gunrock::hip::device_array_t<int32_t> hot_rank_of_device;  // the same on the device (label delivery)
  // this handle IS a slice of a renumbered copy (grx_graph_partition_hot_first): the two arrays
  // above are the permutations grx_partitioned_run translates with
  bool renumbered_slice = false;
  // a renumbered copy: its vertices with edges come first, this many of them (0 = not such a copy)
  unsigned long long leading_connected = 0;

  essentials_amd::graph_type view() const {
    using namespace gunrock;
    auto G = graph::build::from_csr<memory_space_t::device, graph::view_t::csr>(
        n_rows, n_cols, (int32_t)nnz, const_cast<int32_t*>(d_ap), const_cast<int32_t*>(d_aj),
        const_cast<float*>(d_ax));
    if (!max_degree_known) {
      max_degree = essentials_amd::reduce_max_degree(d_ap, n_rows);
      max_degree_known = true;
    }
    // never 0 ("unknown"): an edgeless graph reports 1, which only loosens a sizing bound
    G.properties.max_degree = max_degree ? max_degree : 1ull;
    G.properties.symmetric = symmetry == symmetric;
    G.properties.leading_connected = leading_connected;
    if (in_edges) {
      G.properties.directed = true;
      in_edges->attach_to(G);
    } else if (symmetry == asymmetric) {
      G.properties.directed = true;  // no in-edge view: G.can_pull() is false
    }
    return G;
  }
  void adopt() {
    d_ap = ap.data();
    d_aj = aj.data();
    d_ax = ax.data();
  }
};

namespace essentials_amd {

/// Apply grx_options to the context for the duration of a call.
struct scoped_options {
  gcuda::standard_context_t& ctx;
  gcuda::operator_options_t saved;
  scoped_options(gcuda::standard_context_t& c, const grx_options* opt) : ctx(c), saved(c.options()) {
    if (opt) {
      ctx.options().holes_layout = opt->holes_layout != 0;
      if (opt->hub_threshold > 0)
        ctx.options().hub_threshold = (unsigned)opt->hub_threshold;
      if (opt->chunk_edges > 0)
        ctx.options().chunk_edges = (unsigned)opt->chunk_edges;
      ctx.options().time_kernels = opt->collect_kernel_time != 0;
      ctx.options().chunk_queue_limit = opt->chunk_queue_limit > 0 ? (unsigned long long)opt->chunk_queue_limit : 0ull;
      if (opt->call_every_edge)
        ctx.options().settled_filter = false;
    }
    ctx.kernel_clock().reset();
  }
  ~scoped_options() { ctx.options() = saved; }
};

/// Dispatch a run-time grx_load_balance onto a compile-time load_balance_t.
template <typename F>
int with_load_balance(int lb, F&& f) {
  using operators::load_balance_t;
  switch (lb) {
    case GRX_LB_THREAD_MAPPED: return f(std::integral_constant<load_balance_t, load_balance_t::thread_mapped>());
    case GRX_LB_WARP_MAPPED: return f(std::integral_constant<load_balance_t, load_balance_t::warp_mapped>());
    case GRX_LB_BLOCK_MAPPED: return f(std::integral_constant<load_balance_t, load_balance_t::block_mapped>());
    case GRX_LB_BUCKETING: return f(std::integral_constant<load_balance_t, load_balance_t::bucketing>());
    case GRX_LB_MERGE_PATH:
    case GRX_LB_MERGE_PATH_V2: return f(std::integral_constant<load_balance_t, load_balance_t::merge_path>());
    case GRX_LB_WORK_STEALING: return f(std::integral_constant<load_balance_t, load_balance_t::work_stealing>());
    default: return invalid("unknown load_balance");
  }
}

/// One pass over the labels: reached vertices, the sum of their out-degrees and the source's own
/// degree land in three of the context's device counters (zero between operators).
constexpr int REACH_BLOCK = 1024;
template <typename label_t>
__global__ void __launch_bounds__(REACH_BLOCK)
    reach_stats_kernel(const label_t* labels, label_t unreached, const int32_t* ap, int64_t n,
                       int32_t source, unsigned long long* counters) {
  constexpr int WAVES = REACH_BLOCK / gunrock::hip::wave_size;
  __shared__ unsigned long long s_v[WAVES], s_e[WAVES];
  unsigned long long v = 0, e = 0;
  const int64_t stride = (int64_t)gridDim.x * REACH_BLOCK;
  // four independent label / offset reads in flight per thread
  for (int64_t i0 = blockIdx.x * (int64_t)REACH_BLOCK + threadIdx.x; i0 < n; i0 += 4 * stride) {
    label_t l[4];
    int32_t lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      const bool in = i < n;
      l[k] = in ? labels[i] : unreached;
      lo[k] = in ? ap[i] : 0;
      hi[k] = in ? ap[i + 1] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (l[k] != unreached) {
        ++v;
        e += (unsigned long long)(hi[k] - lo[k]);
      }
  }
  v = gunrock::hip::wave_sum(v);
  e = gunrock::hip::wave_sum(e);
  if ((threadIdx.x & 63) == 0) {
    s_v[threadIdx.x / gunrock::hip::wave_size] = v;
    s_e[threadIdx.x / gunrock::hip::wave_size] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tv = 0, te = 0;
    for (int k = 0; k < WAVES; ++k) {
      tv += s_v[k];
      te += s_e[k];
    }
    if (tv)
      atomicAdd(&counters[gunrock::hip::kernels::C_OUT], tv);
    if (te)
      atomicAdd(&counters[gunrock::hip::kernels::C_WORK], te);
    if (blockIdx.x == 0)
      atomicAdd(&counters[gunrock::hip::kernels::C_SELECT], (unsigned long long)(ap[source + 1] - ap[source]));
  }
}

/// vertices_reached / edges_traversed of a finished traversal and the source's degree (returned),
/// outside the timed region: one kernel + one counters hand-off.
template <typename label_t>
long long reach_stats(grx_graph_s* g, const label_t* d_labels, label_t unreached, int32_t source,
                      gcuda::standard_context_t& ctx, grx_stats* stats) {
  if (!stats)
    return 0;
  const int64_t n = g->n_rows;
  // two 1024-thread workgroups per CU: the three result words take one atomic each per workgroup, and
  // a single device line retires ~90 atomics/us (2048 workgroups made this pass atomic-bound: 57 us)
  const unsigned grid =
      (unsigned)std::min<int64_t>((n + REACH_BLOCK - 1) / REACH_BLOCK, 2 * (int64_t)ctx.compute_units());
  reach_stats_kernel<label_t><<<grid ? grid : 1, REACH_BLOCK, 0, ctx.stream()>>>(d_labels, unreached, g->d_ap, n, source,
                                                                 ctx.workspace().counters());
  GRX_HIP_CHECK(hipGetLastError());
  unsigned long long* m = operators::advance::detail::fetch_counters(ctx);
  stats->vertices_reached = (int64_t)m[hip::kernels::C_OUT];
  stats->edges_traversed = (int64_t)m[hip::kernels::C_WORK];
  return (long long)m[hip::kernels::C_SELECT];
}

}  // namespace essentials_amd
