/**
 * @file reorder.hxx
 * @brief graph::build::hot_first -- the same graph with its vertices renumbered in descending
 * order of out-degree ("hot-first"), built on the device; a data-layout decision of this engine,
 * no counterpart in the reference (its graph views keep the loader's numbering,
 * graph/build.hxx:26-52).
 *
 * Why (measured on MI355X, DESIGN.md section 5 "Round 3"): what an advance pays per edge is the
 * lookup of the destination's label, and what that lookup costs is decided by WHERE the label
 * lives -- 950 G lookups/s out of a CU's L1 / LDS, 260-290 out of the XCD's L2, 60-80 once the
 * array outgrows it (tools/scatter_probe.hip).  On a power-law graph most edges end in few
 * vertices; with those numbered first
 *   - the workgroup-local LDS images of the wide-level kernel (96 KB: a settled bit for the first
 *     786 K ids, or a 2-byte distance bound for the first 48 K) cover the destinations of most
 *     edges whatever numbering the input came with (a Graph500-style scrambled R-MAT-22 runs its
 *     push BFS in 1.58 ms, the generator's own numbering in 1.09 ms, hot-first in 0.92 ms);
 *   - the labels of the remaining hot vertices are contiguous and stay L2-resident.
 * Rows keep their edge order; ties in degree keep the input order (stable sort), so the cold tail
 * stays roughly sequential in both numberings and the result scatter at the end of a run is cheap.
 *
 * The owner keeps both permutations: results of a traversal of the renumbered graph are handed to
 * the caller in ITS numbering (out[vertex_of[r]] = label[r]).
 */
#pragma once

#include <gunrock/graph/graph.hxx>
#include <gunrock/graph/transpose.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/primitives.hxx>

namespace gunrock {
namespace graph {

namespace detail {

template <typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(256)
    degree_keys_kernel(const edge_t* offsets, vertex_t n, unsigned max_degree, unsigned* keys,
                       vertex_t* ids) {
  for (long long v = blockIdx.x * 256ll + threadIdx.x; v < (long long)n; v += (long long)gridDim.x * 256) {
    keys[v] = max_degree - (unsigned)(offsets[v + 1] - offsets[v]);  // ascending key = descending degree
    ids[v] = (vertex_t)v;
  }
}

template <typename vertex_t, typename edge_t>
__global__ void __launch_bounds__(256)
    rank_and_degree_kernel(const vertex_t* vertex_of, const edge_t* offsets, vertex_t n,
                           vertex_t* rank_of, edge_t* new_degree) {
  for (long long r = blockIdx.x * 256ll + threadIdx.x; r <= (long long)n; r += (long long)gridDim.x * 256) {
    if (r == (long long)n) {
      new_degree[r] = 0;
      continue;
    }
    const vertex_t v = vertex_of[r];
    rank_of[v] = (vertex_t)r;
    new_degree[r] = offsets[v + 1] - offsets[v];
  }
}

template <typename vertex_t, typename edge_t, typename weight_t>
__global__ void __launch_bounds__(256)
    renumber_edges_kernel(const edge_t* new_offsets, const vertex_t* row_of_edge,
                          const vertex_t* vertex_of, const vertex_t* rank_of, const edge_t* offsets,
                          const vertex_t* columns, const weight_t* values, long long nnz,
                          vertex_t* new_columns, weight_t* new_values) {
  for (long long p = blockIdx.x * 256ll + threadIdx.x; p < nnz; p += (long long)gridDim.x * 256) {
    const vertex_t r = row_of_edge[p];
    const long long e = (long long)offsets[vertex_of[r]] + (p - (long long)new_offsets[r]);
    new_columns[p] = rank_of[columns[e]];
    new_values[p] = values[e];
  }
}

}  // namespace detail

/// Owning arrays of a renumbered graph and the two permutations.
template <typename vertex_t, typename edge_t, typename weight_t>
struct renumbered_t {
  hip::device_array_t<edge_t> offsets;     // [V + 1]
  hip::device_array_t<vertex_t> indices;   // [E] in the NEW numbering
  hip::device_array_t<weight_t> values;    // [E]
  hip::device_array_t<vertex_t> rank_of;   // [V] new id of an input vertex
  hip::device_array_t<vertex_t> vertex_of; // [V] input id of a new vertex
};

namespace build {

/// Renumber the vertices of a square CSR graph in descending order of out-degree (stable).
template <typename graph_t>
auto hot_first(graph_t& G, gcuda::standard_context_t& ctx, unsigned long long max_degree) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  renumbered_t<vertex_t, edge_t, weight_t> R;
  const std::size_t n = (std::size_t)G.get_number_of_vertices();
  const std::size_t nnz = (std::size_t)G.get_number_of_edges();
  R.offsets.resize(n + 1);
  R.indices.resize(nnz ? nnz : 1);
  R.values.resize(nnz ? nnz : 1);
  R.rank_of.resize(n ? n : 1);
  R.vertex_of.resize(n ? n : 1);
  hipStream_t s = ctx.stream();
  const unsigned grid = (unsigned)ctx.compute_units() * 8;
  if (!n) {
    R.offsets.zero(s);
    GRX_HIP_CHECK(hipStreamSynchronize(s));
    return R;
  }
  // 1. vertices by descending degree, ties in input order
  hip::buffer_t<unsigned> keys(n), keys_sorted(n);
  hip::buffer_t<vertex_t> ids(n);
  detail::degree_keys_kernel<<<grid, 256, 0, s>>>(G.get_row_offsets(), (vertex_t)n, (unsigned)max_degree,
                                                  keys.data(), ids.data());
  unsigned bits = 1;
  while (bits < 32 && (max_degree >> bits))
    ++bits;
  std::size_t bytes = 0;
  GRX_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys.data(), keys_sorted.data(), ids.data(),
                                          R.vertex_of.data(), n, 0, bits, s));
  hip::buffer_t<unsigned char> temp(bytes < 256 ? 256 : bytes);  // never null: rocPRIM would only size
  GRX_HIP_CHECK(rocprim::radix_sort_pairs(temp.data(), bytes, keys.data(), keys_sorted.data(), ids.data(),
                                          R.vertex_of.data(), n, 0, bits, s));
  // 2. inverse permutation, new offsets
  hip::buffer_t<edge_t> degree(n + 1);
  detail::rank_and_degree_kernel<<<grid, 256, 0, s>>>(R.vertex_of.data(), G.get_row_offsets(), (vertex_t)n,
                                                      R.rank_of.data(), degree.data());
  std::size_t scan_bytes = hip::exclusive_sum_temp_bytes(degree.data(), R.offsets.data(), edge_t(0), n + 1);
  hip::buffer_t<unsigned char> scan_temp(scan_bytes < 256 ? 256 : scan_bytes);
  hip::exclusive_sum(scan_temp.data(), scan_bytes, degree.data(), R.offsets.data(), edge_t(0), n + 1, s);
  // 3. rows in the new order, columns through the permutation
  if (nnz) {
    hip::buffer_t<vertex_t> row_of_edge(nnz);
    detail::expand_rows_kernel<<<grid, 256, 0, s>>>(R.offsets.data(), (vertex_t)n, (edge_t)nnz,
                                                    row_of_edge.data());
    detail::renumber_edges_kernel<<<grid, 256, 0, s>>>(
        R.offsets.data(), row_of_edge.data(), R.vertex_of.data(), R.rank_of.data(), G.get_row_offsets(),
        G.get_column_indices(), G.get_nonzero_values(), (long long)nnz, R.indices.data(), R.values.data());
  }
  GRX_HIP_CHECK(hipGetLastError());
  GRX_HIP_CHECK(hipStreamSynchronize(s));
  return R;
}

}  // namespace build
}  // namespace graph
}  // namespace gunrock
