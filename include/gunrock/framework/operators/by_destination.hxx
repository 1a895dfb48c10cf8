/**
 * @file by_destination.hxx
 * @brief The whole-graph advance WITHOUT an output frontier (advance_io_type_t::graph -> none), with
 * the functor called for the edges grouped by DESTINATION.
 *
 * This is the call `pr.hxx` makes once per iteration (reference algorithms/pr.hxx:140-152: for
 * every edge `atomic::add(&p[dst], plast[src] * iweights[src] * w)`), and the reference runs it
 * row by row (advance/block_mapped.hxx:116-146 over all vertices).  Row by row, 268 M float
 * atomics of a directed R-MAT-24 retire at 6.5 G/s -- 41 ms an iteration, 1 % of the HBM roofline
 * -- because the hottest destinations receive hundreds of thousands of them and ONE 128-byte line
 * retires ~90 read-modify-writes per microsecond however many CUs queue for it (DESIGN.md section
 * 5, tools/pr_push_probe.py, tools/pr_hot_probe.py).
 *
 * An advance promises the functor ONE call per edge with that edge's (source, destination, edge
 * id, weight); it promises no order.  So from the second such call on the same graph this
 * operator walks a copy of the edge list sorted by destination (16 bytes per edge, one coalesced
 * load per lane, perfectly balanced: no rows, no hubs), where the lanes of a wave that hold edges
 * into the same vertex are neighbours -- and `math::atomic::add` (util/math.hxx: add_runs) sends
 * ONE read-modify-write per run of neighbouring lanes with the same address.  The unchanged
 * `pr.hxx` then issues ~one atomic per destination and wave instead of one per edge.
 *
 * The sorted copy is the engine's (built on the device: stable radix sort of the edge ids by
 * column, sources by a gather), kept in the context's workspace and identified by the addresses
 * of the CSR arrays AND a 64-bit fingerprint of their contents, checked once per enactor: memory
 * reused for another graph never matches (0.4 ms per run on R-MAT-24).
 */
#pragma once

#include <gunrock/graph/graph.hxx>
#include <gunrock/graph/transpose.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/primitives.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace by_destination {

template <typename vertex_t, typename edge_t, typename weight_t>
struct alignas(16) item_t {
  vertex_t source;
  vertex_t destination;
  edge_t edge;
  weight_t weight;
};

namespace k {

constexpr unsigned BLOCK = 256;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

/// sum over the 4-byte words w[i] of mix64(salt + i, w[i]): position-dependent, order-free to add up
template <typename word_t>
__global__ void __launch_bounds__(BLOCK)
    fingerprint_kernel(const word_t* words, long long n, unsigned long long salt,
                       unsigned long long* sum) {
  unsigned long long h = 0;
  for (long long i = blockIdx.x * (long long)BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK)
    h += mix64(((salt + (unsigned long long)i) << 32) ^ words[i]);
  for (int d = 32; d; d >>= 1)
    h += __shfl_xor(h, d);
  if ((threadIdx.x & 63) == 0)
    ::atomicAdd(sum, h);
}

template <typename item_type, typename vertex_t, typename edge_t, typename weight_t>
__global__ void __launch_bounds__(BLOCK)
    pack_kernel(const vertex_t* rows, const vertex_t* sorted_columns, const edge_t* sorted_edges,
                const weight_t* values, long long n, item_type* items) {
  for (long long j = blockIdx.x * (long long)BLOCK + threadIdx.x; j < n; j += (long long)gridDim.x * BLOCK) {
    const edge_t e = sorted_edges[j];
    item_type it;
    it.source = rows[e];
    it.destination = sorted_columns[j];
    it.edge = e;
    it.weight = values[e];
    items[j] = it;
  }
}

/// the list is read once per walk and is far larger than any cache: a streaming (non-temporal)
/// load keeps it from evicting what the functor gathers from
template <bool streaming, typename item_type>
__device__ __forceinline__ item_type load_item(const item_type* p) {
  if constexpr (streaming && sizeof(item_type) == 16) {
    using words_t = unsigned __attribute__((ext_vector_type(4)));
    const words_t raw = __builtin_nontemporal_load(reinterpret_cast<const words_t*>(p));
    item_type it;
    __builtin_memcpy(&it, &raw, sizeof it);
    return it;
  } else {
    return *p;
  }
}

/// one edge per lane, neighbouring lanes neighbouring positions of the destination-sorted list
template <bool streaming, typename item_type, typename operator_t>
__global__ void __launch_bounds__(BLOCK)
    expand_kernel(const item_type* __restrict__ items, long long n, operator_t op) {
  for (long long j = blockIdx.x * (long long)BLOCK + threadIdx.x; j < n; j += (long long)gridDim.x * BLOCK) {
    const item_type it = load_item<streaming>(items + j);
    op(it.source, it.destination, it.edge, it.weight);
  }
}

}  // namespace k

namespace detail {

template <typename graph_t>
unsigned long long fingerprint(graph_t& G, gcuda::standard_context_t& context) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  static_assert(sizeof(vertex_t) % 4 == 0 && sizeof(edge_t) % 4 == 0 && sizeof(weight_t) % 4 == 0,
                "fingerprint reads 4-byte words");
  const long long n = (long long)G.get_number_of_vertices(), nnz = (long long)G.get_number_of_edges();
  auto* sum = reinterpret_cast<unsigned long long*>(context.workspace().scratch(256));
  hipStream_t s = context.stream();
  GRX_HIP_CHECK(hipMemsetAsync(sum, 0, sizeof(unsigned long long), s));
  const unsigned grid = (unsigned)context.compute_units() * 8u;
  auto pass = [&](const void* p, long long words, unsigned long long salt) {
    if (words > 0)
      k::fingerprint_kernel<unsigned><<<grid, k::BLOCK, 0, s>>>(reinterpret_cast<const unsigned*>(p), words, salt, sum);
  };
  pass(G.get_row_offsets(), (n + 1) * (long long)(sizeof(edge_t) / 4), 1ull << 28);
  pass(G.get_column_indices(), nnz * (long long)(sizeof(vertex_t) / 4), 2ull << 28);
  pass(G.get_nonzero_values(), nnz * (long long)(sizeof(weight_t) / 4), 3ull << 28);
  GRX_HIP_CHECK(hipGetLastError());
  unsigned long long h = 0;
  GRX_HIP_CHECK(hipMemcpyAsync(&h, sum, sizeof h, hipMemcpyDeviceToHost, s));
  GRX_HIP_CHECK(hipStreamSynchronize(s));
  return h ? h : 1;
}

/// Sort the edge list by destination into `cache.items`; false when device memory does not allow.
template <typename graph_t>
bool build(graph_t& G, gcuda::workspace_t::by_destination_t& cache, gcuda::standard_context_t& context) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  using item_type = item_t<vertex_t, edge_t, weight_t>;
  const std::size_t n = (std::size_t)G.get_number_of_vertices();
  const std::size_t nnz = (std::size_t)G.get_number_of_edges();
  std::size_t free_bytes = 0, total_bytes = 0;
  GRX_HIP_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
  // the list itself + {edge ids, sorted ids, sorted columns, rows} + the sort's own storage
  const std::size_t need = nnz * (sizeof(item_type) + 2 * sizeof(edge_t) + 2 * sizeof(vertex_t)) +
                           nnz * (sizeof(edge_t) + sizeof(vertex_t)) + (64u << 20);
  if (need > free_bytes + cache.items.capacity())
    return false;
  hipStream_t s = context.stream();
  const unsigned grid = (unsigned)context.compute_units() * 8u;
  // gigabytes used once: straight back to the device afterwards, not parked for reuse
  hip::buffer_t<edge_t> ids, sorted_edges;
  hip::buffer_t<vertex_t> sorted_columns, rows;
  hip::buffer_t<unsigned char> temp;
  for (auto* b : {&ids, &sorted_edges})
    b->set_parking(false);
  for (auto* b : {&sorted_columns, &rows})
    b->set_parking(false);
  temp.set_parking(false);
  cache.items.set_parking(false);
  std::size_t bytes = 0;
  unsigned bits = 1;
  while (bits < 8 * sizeof(vertex_t) && (n >> bits))
    ++bits;
  try {
    cache.items.reserve(nnz * sizeof(item_type));
    ids.reserve(nnz);
    sorted_edges.reserve(nnz);
    sorted_columns.reserve(nnz);
    rows.reserve(nnz);
    GRX_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, G.get_column_indices(), sorted_columns.data(),
                                            ids.data(), sorted_edges.data(), nnz, 0, bits, s));
    temp.reserve(bytes < 256 ? 256 : bytes);  // never null: that would be a size query
  } catch (const error::exception_t&) {
    // no room after all (memory held elsewhere in the process): walk row by row, as without the list
    (void)hipGetLastError();
    cache.items.release();
    return false;
  }
  hip::for_each_index_on(nnz, ids.data(), s);
  {
    GRX_HIP_CHECK(rocprim::radix_sort_pairs(temp.data(), bytes, G.get_column_indices(),
                                            sorted_columns.data(), ids.data(), sorted_edges.data(), nnz, 0,
                                            bits, s));
    graph::detail::expand_rows_kernel<<<grid, 256, 0, s>>>(G.get_row_offsets(), (vertex_t)n, (edge_t)nnz,
                                                           rows.data());
    k::pack_kernel<<<grid, k::BLOCK, 0, s>>>(rows.data(), sorted_columns.data(), sorted_edges.data(),
                                             G.get_nonzero_values(), (long long)nnz,
                                             reinterpret_cast<item_type*>(cache.items.data()));
    GRX_HIP_CHECK(hipGetLastError());
    GRX_HIP_CHECK(hipStreamSynchronize(s));
  }
  return true;
}

}  // namespace detail

/**
 * @brief The destination-sorted edge list of G if this call should walk it, else nullptr (the
 * caller then expands row by row).  `run_id`: bsp.hxx's enactor id, 0 outside an enactor (the
 * fingerprint is then compared on every call).
 */
template <typename graph_t>
const void* prepared(graph_t& G, unsigned long long run_id, gcuda::standard_context_t& context) {
  const unsigned long long min_edges = context.options().by_destination_min_edges;
  const std::size_t nnz = (std::size_t)G.get_number_of_edges();
  if (!min_edges || nnz < min_edges)
    return nullptr;
  auto& cache = context.workspace().by_destination();
  const bool same_place = cache.offsets == (const void*)G.get_row_offsets() &&
                          cache.indices == (const void*)G.get_column_indices() &&
                          cache.values == (const void*)G.get_nonzero_values() &&
                          cache.vertices == (std::size_t)G.get_number_of_vertices() && cache.edges == nnz;
  if (same_place && run_id && cache.checked_for == run_id && (cache.built || cache.refused))
    return cache.built ? cache.items.data() : nullptr;  // the same enactor compared the contents already
  // an enactor that advances over several graphs in turn would drop and fingerprint a list per call
  if (run_id && cache.alternating_in == run_id && cache.switches >= 3)
    return nullptr;
  const unsigned long long h = detail::fingerprint(G, context);
  if (!same_place || cache.fingerprint != h) {
    if (run_id && cache.checked_for == run_id) {  // this enactor was on another graph a call ago
      cache.switches = cache.alternating_in == run_id ? cache.switches + 1 : 1;
      cache.alternating_in = run_id;
    }
    cache.offsets = G.get_row_offsets();
    cache.indices = G.get_column_indices();
    cache.values = G.get_nonzero_values();
    cache.vertices = (std::size_t)G.get_number_of_vertices();
    cache.edges = nnz;
    cache.fingerprint = h;
    cache.calls = 0;
    cache.built = false;
    cache.refused = false;
    cache.items.release();  // the other graph's list
  }
  cache.checked_for = run_id;
  if (cache.refused)
    return nullptr;
  if (!cache.built) {
    if (++cache.calls < 2)
      return nullptr;  // a graph walked once is not worth a sort
    if (!detail::build(G, cache, context)) {
      cache.refused = true;
      return nullptr;
    }
    cache.built = true;
  }
  return cache.items.data();
}

template <typename graph_t, typename operator_t>
void enqueue(graph_t& G, const void* items, operator_t op, gcuda::standard_context_t& context) {
  using item_type = item_t<typename graph_t::vertex_type, typename graph_t::edge_type,
                           typename graph_t::weight_type>;
  const long long nnz = (long long)G.get_number_of_edges();
  const long long blocks = (nnz + k::BLOCK - 1) / k::BLOCK;
  const long long resident = (long long)context.compute_units() * 32;
  const unsigned grid = (unsigned)(blocks < resident ? blocks : resident);
  const char* e = std::getenv("GRX_BY_DESTINATION_STREAM");
  if (e && std::atoi(e) == 0)
    k::expand_kernel<false><<<grid, k::BLOCK, 0, context.stream()>>>(
        reinterpret_cast<const item_type*>(items), nnz, op);
  else
    k::expand_kernel<true><<<grid, k::BLOCK, 0, context.stream()>>>(
        reinterpret_cast<const item_type*>(items), nnz, op);
  GRX_HIP_CHECK(hipGetLastError());
}

}  // namespace by_destination
}  // namespace advance
}  // namespace operators
}  // namespace gunrock
