/**
 * @file transpose.hxx
 * @brief graph::build::transpose -- the in-edge (CSC) arrays of a CSR graph, built on the device.
 *
 * The reference builds its csc view by sorting the CSR's column_indices IN PLACE together with
 * row indices and values (graph/detail/build.hxx:96-113), which is why it refuses csr and csc
 * views together.  Here the transpose is a separate, owning object (stable rocPRIM radix sort by
 * column; in-neighbours of a vertex appear in ascending source order), attached to the graph
 * view with attach_in_edges(); the CSR stays untouched.  Needed by pull advances on DIRECTED
 * graphs (an undirected CSR is its own transpose).
 */
#pragma once

#include <gunrock/graph/graph.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/primitives.hxx>

namespace gunrock {
namespace graph {

namespace detail {

template <typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(256)
    expand_rows_kernel(const edge_t* offsets, vertex_t n_rows, edge_t nnz, vertex_t* rows) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < (long long)nnz;
       e += (long long)gridDim.x * 256) {
    vertex_t lo = 0, hi = n_rows;  // offsets[lo] <= e < offsets[hi]
    while (hi - lo > 1) {
      const vertex_t mid = lo + (hi - lo) / 2;
      if ((long long)offsets[mid] <= e)
        lo = mid;
      else
        hi = mid;
    }
    rows[e] = lo;
  }
}

template <typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(256)
    offsets_from_sorted_kernel(const vertex_t* sorted_keys, long long total, vertex_t n_keys,
                               edge_t* offsets) {
  // offsets[r] = first position whose key is >= r, r in [0, n_keys]
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i <= total; i += (long long)gridDim.x * 256) {
    const long long hi = (i < total) ? (long long)sorted_keys[i] : (long long)n_keys;
    const long long lo = (i == 0) ? 0 : (long long)sorted_keys[i - 1] + 1;
    for (long long r = lo; r <= hi && r <= (long long)n_keys; ++r)
      offsets[r] = (edge_t)i;
  }
}

template <typename weight_t, typename edge_t>
__global__ void __launch_bounds__(256)
    gather_kernel(const weight_t* src, const edge_t* perm, long long n, weight_t* dst) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    dst[i] = src[perm[i]];
}

}  // namespace detail

/// Owning in-edge arrays of a graph.
template <typename vertex_t, typename edge_t, typename weight_t>
struct transposed_t {
  hip::device_array_t<edge_t> offsets;    // [V + 1]
  hip::device_array_t<vertex_t> indices;  // [E] source of every in-edge, grouped by destination
  hip::device_array_t<weight_t> values;   // [E]
  hip::device_array_t<edge_t> edge_ids;   // [E] position of the in-edge in the original CSR

  template <typename graph_t>
  void attach_to(graph_t& G) {
    G.attach_in_edges(offsets.data(), indices.data(), values.data());
  }
};

namespace build {

template <typename graph_t>
auto transpose(graph_t& G, gcuda::standard_context_t& ctx) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  transposed_t<vertex_t, edge_t, weight_t> T;
  const std::size_t n = (std::size_t)G.get_number_of_vertices();
  const std::size_t nnz = (std::size_t)G.get_number_of_edges();
  T.offsets.resize(n + 1);
  T.indices.resize(nnz ? nnz : 1);
  T.values.resize(nnz ? nnz : 1);
  T.edge_ids.resize(nnz ? nnz : 1);
  hipStream_t s = ctx.stream();
  const unsigned grid = (unsigned)ctx.compute_units() * 8;
  if (nnz) {
    hip::buffer_t<vertex_t> cols_sorted(nnz);
    hip::buffer_t<edge_t> ids(nnz);
    // edge ids 0..nnz-1, sorted stably by destination
    hip::for_each_index_on(nnz, ids.data(), s);
    std::size_t bytes = 0;
    GRX_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, G.get_column_indices(), cols_sorted.data(),
                                            ids.data(), T.edge_ids.data(), nnz, 0,
                                            8 * sizeof(vertex_t), s));
    // never a null temp: rocPRIM reads "temp == nullptr" as a size query and sorts nothing
    hip::buffer_t<unsigned char> temp(bytes < 256 ? 256 : bytes);
    GRX_HIP_CHECK(rocprim::radix_sort_pairs(temp.data(), bytes, G.get_column_indices(),
                                            cols_sorted.data(), ids.data(), T.edge_ids.data(), nnz, 0,
                                            8 * sizeof(vertex_t), s));
    // in-neighbour = row of the original edge; weight follows the edge
    hip::buffer_t<vertex_t> rows(nnz);
    detail::expand_rows_kernel<<<grid, 256, 0, s>>>(G.get_row_offsets(), (vertex_t)n, (edge_t)nnz,
                                                    rows.data());
    detail::gather_kernel<<<grid, 256, 0, s>>>(rows.data(), T.edge_ids.data(), (long long)nnz,
                                               T.indices.data());
    detail::gather_kernel<<<grid, 256, 0, s>>>(G.get_nonzero_values(), T.edge_ids.data(),
                                               (long long)nnz, T.values.data());
    detail::offsets_from_sorted_kernel<<<grid, 256, 0, s>>>(cols_sorted.data(), (long long)nnz,
                                                            (vertex_t)n, T.offsets.data());
    GRX_HIP_CHECK(hipGetLastError());
    GRX_HIP_CHECK(hipStreamSynchronize(s));
  } else {
    T.offsets.zero(s);
    GRX_HIP_CHECK(hipStreamSynchronize(s));
  }
  return T;
}

}  // namespace build
}  // namespace graph
}  // namespace gunrock
