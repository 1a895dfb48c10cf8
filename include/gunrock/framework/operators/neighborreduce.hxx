/**
 * @file neighborreduce.hxx
 * @brief operators::neighborreduce::execute -- per-vertex reduction over its out-edges:
 *        output[v] = reduce(arithmetic_op, init_value, { op(e) : e an out-edge of v }).
 *
 * API of reference framework/operators/neighborreduce/neighborreduce.hxx:55-101 (a wrapper around
 * ModernGPU's transform_segreduce there); its one in-tree user is the pull form of SpMV
 * (algorithms/spmv.hxx:107-128: y = A x without atomics).  Own kernels, no ModernGPU:
 *   - rows shorter than NR_HUB edges: NR_GROUP (16) lanes of a wavefront share one row (an R-MAT
 *     row has ~16 edges: one 64-B column segment per step), the lanes' partials are combined by a
 *     fixed butterfly, four rows per wavefront, persistent workgroups;
 *   - longer rows are appended to a list (one wavefront-aggregated atomic) and a second kernel gives
 *     each a whole workgroup: a strided walk, then a fixed tree over wavefronts and lanes.
 * The combination order is a function of the row's length only: results are reproducible for any
 * arithmetic_op (float sums included), unlike atomics.  `arithmetic_op` must be associative and
 * commutative up to what the client accepts from a segmented reduction.
 */
#pragma once

#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/framework/operators/advance.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/primitives.hxx>

namespace gunrock {
namespace operators {
namespace neighborreduce {

namespace detail {

constexpr int NR_BLOCK = 256;
constexpr int NR_GROUP = 16;
constexpr unsigned NR_HUB = 512;

template <typename graph_t, typename output_t, typename operator_t, typename arithmetic_t>
__global__ void __launch_bounds__(NR_BLOCK)
    row_group_reduce_kernel(graph_t G, output_t* __restrict__ output, operator_t op,
                            arithmetic_t arithmetic_op, output_t init_value,
                            typename graph_t::vertex_type* __restrict__ hub_rows,
                            unsigned long long* hub_count) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  const std::size_t n = (std::size_t)G.get_number_of_vertices();
  const int lane = threadIdx.x & (NR_GROUP - 1);
  const std::size_t groups = (std::size_t)gridDim.x * (NR_BLOCK / NR_GROUP);
  const std::size_t trips = (n + groups - 1) / groups;  // wave-uniform: shuffles below
  std::size_t row = ((std::size_t)blockIdx.x * NR_BLOCK + threadIdx.x) / NR_GROUP;
  for (std::size_t t = 0; t < trips; ++t, row += groups) {
    output_t acc = init_value;
    bool any = false, hub = false;
    if (row < n) {
      const edge_t first = G.get_starting_edge((vertex_t)row);
      const unsigned deg = (unsigned)(G.get_starting_edge((vertex_t)(row + 1)) - first);
      hub = deg >= NR_HUB;
      if (!hub)
        for (unsigned j = lane; j < deg; j += NR_GROUP) {
          const output_t x = op(first + (edge_t)j);
          acc = any ? arithmetic_op(acc, x) : arithmetic_op(init_value, x);
          any = true;
        }
    }
    // butterfly over the 16 lanes of the group; lanes without an element carry init_value
#pragma unroll
    for (int d = NR_GROUP / 2; d > 0; d >>= 1) {
      const output_t other = __shfl_xor(acc, d, NR_GROUP);
      acc = arithmetic_op(acc, other);
    }
    if (row < n && lane == 0 && !hub)
      output[row] = acc;
    // hub rows: one entry per row in the list
    const bool lists = hub && lane == 0;
    const unsigned long long m = __ballot(lists);
    if (m) {
      unsigned long long base = 0;
      if (hip::lane_id() == 0)
        base = atomicAdd(hub_count, (unsigned long long)__popcll(m));
      base = __shfl(base, 0, hip::wave_size);
      if (lists)
        hub_rows[base + hip::rank_in_mask(m)] = (vertex_t)row;
    }
  }
}

template <typename graph_t, typename output_t, typename operator_t, typename arithmetic_t>
__global__ void __launch_bounds__(NR_BLOCK)
    hub_row_reduce_kernel(graph_t G, output_t* __restrict__ output, operator_t op,
                          arithmetic_t arithmetic_op, output_t init_value,
                          const typename graph_t::vertex_type* __restrict__ hub_rows,
                          const unsigned long long* hub_count) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  __shared__ output_t s_part[NR_BLOCK];
  const unsigned long long n_hubs =
      __hip_atomic_load(hub_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned long long h = blockIdx.x; h < n_hubs; h += gridDim.x) {
    const vertex_t row = hub_rows[h];
    const edge_t first = G.get_starting_edge(row);
    const unsigned deg = (unsigned)(G.get_starting_edge(row + 1) - first);
    output_t acc = init_value;
    for (unsigned j = threadIdx.x; j < deg; j += NR_BLOCK)
      acc = arithmetic_op(acc, op(first + (edge_t)j));
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int d = NR_BLOCK / 2; d > 0; d >>= 1) {  // fixed tree: reproducible
      if ((int)threadIdx.x < d)
        s_part[threadIdx.x] = arithmetic_op(s_part[threadIdx.x], s_part[threadIdx.x + d]);
      __syncthreads();
    }
    if (threadIdx.x == 0)
      output[row] = s_part[0];
    __syncthreads();  // s_part is rewritten by the next hub
  }
}

}  // namespace detail

/**
 * @brief output[v] = reduction of op(e) over the out-edges e of every vertex v (whole graph).
 * `init_value` must be the identity of `arithmetic_op` (it is what an edgeless vertex receives and
 * what idle lanes contribute).  Synchronous, like every operator.
 */
template <advance_io_type_t input_t = advance_io_type_t::graph,
          typename graph_t,
          typename enactor_t,
          typename output_t,
          typename operator_t,
          typename arithmetic_t>
void execute(graph_t& G,
             enactor_t* E,
             output_t* output,
             operator_t op,
             arithmetic_t arithmetic_op,
             output_t init_value,
             gcuda::multi_context_t& context) {
  (void)E;
  using vertex_t = typename graph_t::vertex_type;
  error::throw_if_exception(context.size() != 1, "`context.size() != 1` not supported");
  error::throw_if_exception(input_t != advance_io_type_t::graph,
                            "neighborreduce: only the whole graph as input is supported");
  auto& ctx = *context.get_context(0);
  const std::size_t n = (std::size_t)G.get_number_of_vertices();
  if (n == 0)
    return;
  auto& ws = ctx.workspace();
  // scratch: [hub count | hub rows]
  auto* count = reinterpret_cast<unsigned long long*>(ws.scratch(16 + n * sizeof(vertex_t)));
  auto* hubs = reinterpret_cast<vertex_t*>(count + 2);
  GRX_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(unsigned long long), ctx.stream()));
  const unsigned grid = (unsigned)ctx.compute_units() * 8u;
  detail::row_group_reduce_kernel<<<grid, detail::NR_BLOCK, 0, ctx.stream()>>>(
      G, output, op, arithmetic_op, init_value, hubs, count);
  detail::hub_row_reduce_kernel<<<grid, detail::NR_BLOCK, 0, ctx.stream()>>>(
      G, output, op, arithmetic_op, init_value, hubs, count);
  GRX_HIP_CHECK(hipGetLastError());
  ctx.synchronize();
}

}  // namespace neighborreduce
}  // namespace operators
}  // namespace gunrock
